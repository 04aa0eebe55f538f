"""StridedView -- host-side mirror of StridedViews.jl's `StridedView{T,N,A,F}` as consumed by the
reference (5-field constructor call /root/reference/src/broadcast.jl:64; field reads
src/macros.jl:36-38, src/linalg.jl:51,66-68).

A view is (parent, size, strides, offset, op): element (i_1..i_N) lives at
parent[offset + sum_k i_k*strides[k]] (0-based here; the reference is 1-based) and `op` in
{identity, conj} is applied lazily on load and store.  Strides are in elements and may be 0
(broadcast) or negative (reversed ranges).  All view algebra below is lazy -- it only rewrites
(size, strides, offset, op) -- exactly like `permutedims`/`adjoint`/`sreshape`/`sview` in the
reference's dependency.  The parent is a torch tensor on a HIP device (the product path) or a
NumPy array (host views, used by the test-suite together with the CPU oracle).
"""
from __future__ import annotations

import builtins

import numpy as np

from . import _lib as L

builtins_sum = builtins.sum

_NP2SMR = {
    np.dtype(np.float32): L.SMR_F32, np.dtype(np.float64): L.SMR_F64,
    np.dtype(np.complex64): L.SMR_C32, np.dtype(np.complex128): L.SMR_C64,
    np.dtype(np.int8): L.SMR_I8, np.dtype(np.int16): L.SMR_I16, np.dtype(np.int32): L.SMR_I32,
    np.dtype(np.int64): L.SMR_I64, np.dtype(np.uint8): L.SMR_U8, np.dtype(np.bool_): L.SMR_BOOL,
    np.dtype(np.uint16): L.SMR_U16, np.dtype(np.uint32): L.SMR_U32, np.dtype(np.uint64): L.SMR_U64,
}


def smr_dtype(dt) -> int:
    dt = np.dtype(dt)
    if dt not in _NP2SMR:
        raise TypeError(f"element type {dt} is not supported on the device")
    return _NP2SMR[dt]


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _torch_np_dtype(t):
    import torch
    table = {
        torch.float32: np.float32, torch.float64: np.float64, torch.complex64: np.complex64,
        torch.complex128: np.complex128, torch.int8: np.int8, torch.int16: np.int16,
        torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8, torch.bool: np.bool_,
    }
    for name, npdt in (("uint16", np.uint16), ("uint32", np.uint32), ("uint64", np.uint64)):  # torch >= 2.3
        if hasattr(torch, name):
            table[getattr(torch, name)] = npdt
    if t.dtype not in table:
        raise TypeError(f"torch dtype {t.dtype} is not supported")
    return np.dtype(table[t.dtype])


def _np_torch_dtype(dt):
    import torch
    table = {
        np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
        np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128,
        np.dtype(np.int8): torch.int8, np.dtype(np.int16): torch.int16, np.dtype(np.int32): torch.int32,
        np.dtype(np.int64): torch.int64, np.dtype(np.uint8): torch.uint8, np.dtype(np.bool_): torch.bool,
    }
    for name, npdt in (("uint16", np.uint16), ("uint32", np.uint32), ("uint64", np.uint64)):
        if hasattr(torch, name):
            table[np.dtype(npdt)] = getattr(torch, name)
    return table[np.dtype(dt)]


class DimensionMismatch(ValueError):
    """Julia's DimensionMismatch (thrown before the engine runs: src/mapreduce.jl:43-46)."""


def _reshape_strides(newsize, oldsize, oldstrides):
    """Strides of a column-major reshape that needs no copy, or raise (sreshape semantics:
    old dims may only be merged when they are jointly contiguous)."""
    if int(np.prod(newsize, dtype=np.int64)) != int(np.prod(oldsize, dtype=np.int64)):
        raise DimensionMismatch(f"cannot reshape {tuple(oldsize)} to {tuple(newsize)}")
    old = [(d, s) for d, s in zip(oldsize, oldstrides) if d != 1]
    new = list(newsize)
    out = [0] * len(new)
    ni = oi = 0
    last = 1
    while ni < len(new):
        if new[ni] == 1:
            out[ni] = last
            ni += 1
            continue
        if oi >= len(old):
            raise DimensionMismatch("reshape mismatch")
        np_, op_ = new[ni], old[oi][0]
        nj, oj = ni + 1, oi + 1
        while np_ != op_:
            if np_ < op_:
                if nj >= len(new):
                    raise DimensionMismatch("reshape mismatch")
                np_ *= new[nj]
                nj += 1
            else:
                if oj >= len(old):
                    raise DimensionMismatch("reshape mismatch")
                op_ *= old[oj][0]
                oj += 1
        for k in range(oi, oj - 1):
            if old[k + 1][1] != old[k][0] * old[k][1]:
                raise ValueError("sreshape: dims are not contiguous, a strided reshape is impossible")
        s = old[oi][1]
        for k in range(ni, nj):
            out[k] = s
            s *= new[k]
        last = s
        ni, oi = nj, oj
    return tuple(out)


class StridedView:
    __array_priority__ = 1000

    def __init__(self, parent, size=None, strides=None, offset=0, op="identity"):
        if isinstance(parent, StridedView):
            p = parent
            parent, size0, strides0, offset0, op0 = p.parent, p.size, p.strides, p.offset, p.op
            if size is None:
                size, strides, offset, op = size0, strides0, offset0, op0
        self.parent = parent
        if _is_torch(parent):
            self.dtype = _torch_np_dtype(parent)
            self._base = int(parent.data_ptr())
            self._device = parent.device
            psize, pstrides = tuple(parent.shape), tuple(parent.stride())
        elif isinstance(parent, np.ndarray):
            self.dtype = parent.dtype
            self._base = int(parent.ctypes.data)
            self._device = None
            isz = parent.dtype.itemsize
            psize = tuple(parent.shape)
            pstrides = tuple(s // isz for s in parent.strides)
            if any(s % isz for s in parent.strides):
                raise ValueError("parent strides are not multiples of the element size")
        else:
            raise TypeError("StridedView parent must be a torch tensor or a numpy array")
        if size is None:
            size, strides = psize, pstrides
        self.size = tuple(int(d) for d in size)
        self.strides = tuple(int(s) for s in strides)
        if len(self.size) != len(self.strides):
            raise ValueError("size/strides rank mismatch")
        self.offset = int(offset)
        if op not in ("identity", "conj"):
            raise ValueError("op must be 'identity' or 'conj'")
        # conj of a real view is the view itself (Base.conj!(a::StridedView{<:Real}) = a)
        self.op = op if np.issubdtype(self.dtype, np.complexfloating) else "identity"

    # ---- basic queries -----------------------------------------------------------------------
    @property
    def ndim(self):
        return len(self.size)

    @property
    def shape(self):
        return self.size

    def __len__(self):
        return int(np.prod(self.size, dtype=np.int64)) if self.size else 1

    length = __len__

    @property
    def on_device(self) -> bool:
        return self._device is not None and self._device.type == "cuda"

    def stride(self, d):
        return self.strides[d] if d < self.ndim else 1

    def _size(self, d):
        return self.size[d] if d < self.ndim else 1

    def _with(self, size, strides, offset=None, op=None):
        return StridedView(self.parent, size, strides, self.offset if offset is None else offset,
                           self.op if op is None else op)

    def __repr__(self):
        where = f"device={self._device}" if self._device is not None else "host"
        return (f"StridedView({self.dtype}, size={self.size}, strides={self.strides}, "
                f"offset={self.offset}, op={self.op}, {where})")

    # ---- lazy view algebra (StridedViews.jl) ---------------------------------------------------
    def permutedims(self, p):
        p = tuple(int(i) for i in p)
        if sorted(p) != list(range(self.ndim)):
            raise ValueError(f"{p} is not a permutation of 0..{self.ndim - 1}")
        return self._with(tuple(self.size[i] for i in p), tuple(self.strides[i] for i in p))

    def transpose(self):
        if self.ndim != 2:
            raise ValueError("transpose needs a 2-d view")
        return self.permutedims((1, 0))

    def conj(self):
        return self._with(self.size, self.strides, op="conj" if self.op == "identity" else "identity")

    def adjoint(self):
        return self.transpose().conj()

    @property
    def T(self):
        return self.transpose()

    @property
    def H(self):
        return self.adjoint()

    def sreshape(self, newsize):
        newsize = tuple(int(d) for d in newsize)
        return self._with(newsize, _reshape_strides(newsize, self.size, self.strides))

    def sview(self, *idx):
        """`sview(a, I...)`: ranges keep a dim (stride scaled by the step, negative steps
        allowed), integers drop it.  0-based, Python slice semantics."""
        if len(idx) == 1 and isinstance(idx[0], tuple):
            idx = idx[0]
        idx = list(idx)
        if any(i is Ellipsis for i in idx):
            k = idx.index(Ellipsis)
            idx = idx[:k] + [slice(None)] * (self.ndim - len(idx) + 1) + idx[k + 1:]
        if len(idx) != self.ndim:
            raise IndexError(f"need {self.ndim} indices, got {len(idx)}")
        size, strides, off = [], [], self.offset
        for d, i in enumerate(idx):
            n, s = self.size[d], self.strides[d]
            if isinstance(i, slice):
                r = range(*i.indices(n))
                off += (r.start if len(r) else 0) * s
                size.append(len(r))
                strides.append(s * r.step)
            else:
                i = int(i)
                if i < 0:
                    i += n
                if not 0 <= i < n:
                    raise IndexError("index out of range")
                off += i * s
        return self._with(tuple(size), tuple(strides), offset=off)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        v = self.sview(*idx)
        if v.ndim == 0:
            return v.item()
        return v

    def item(self):
        if len(self) != 1:
            raise ValueError("item() needs a one-element view")
        return self.toarray().reshape(-1)[0].item()

    # ---- allocation / materialisation ----------------------------------------------------------
    def similar(self, dtype=None, size=None):
        """`similar(a, T, dims)`: fresh dense column-major array on the same memory space."""
        dtype = np.dtype(self.dtype if dtype is None else dtype)
        size = self.size if size is None else tuple(int(d) for d in size)
        n = int(np.prod(size, dtype=np.int64)) if size else 1
        cm = []
        s = 1
        for d in size:
            cm.append(s)
            s *= d
        if _is_torch(self.parent):
            import torch
            buf = torch.empty(max(n, 1), dtype=_np_torch_dtype(dtype), device=self.parent.device)
        else:
            buf = np.empty(max(n, 1), dtype=dtype)
        return StridedView(buf, size, tuple(cm), 0, "identity")

    def toarray(self) -> np.ndarray:
        """Host copy of the logical array (applies op).  For tests and small results; the bulk
        device->host path is `Array(view)` in mapreduce.py."""
        if _is_torch(self.parent):
            import ctypes
            import importlib
            import torch
            # inside `with S.Stream():` the launches went to a library-owned stream, which torch's streams are not ordered against:
            # finish it before torch gathers (found by tools/fuzz_more.py OWN_STREAM=1, whose harness read stale results)
            own = importlib.import_module(".mapreduce", __package__)._STREAM_OVERRIDE.stack
            if own:
                L.check(L.load().smr_stream_sync(ctypes.c_void_p(own[-1])))
            p = self.parent.detach()
            st = p.untyped_storage()
            n_after = st.nbytes() // self.dtype.itemsize - p.storage_offset()
            flat_t = torch.empty(0, dtype=p.dtype, device=p.device).set_(st, p.storage_offset(), (n_after,), (1,))
            idx = np.full(self.size, self.offset, dtype=np.int64)
            for d, (n, s) in enumerate(zip(self.size, self.strides)):
                shp = [1] * self.ndim
                shp[d] = n
                idx = idx + (np.arange(n, dtype=np.int64) * s).reshape(shp)
            if self.dtype.kind == "u" and self.dtype.itemsize > 1:
                # torch has no gather for the wide unsigned types: index the same bits as signed integers
                sdt = {2: torch.int16, 4: torch.int32, 8: torch.int64}[self.dtype.itemsize]
                out = flat_t.view(sdt)[torch.from_numpy(np.ascontiguousarray(idx)).to(p.device)].cpu().numpy().view(self.dtype)
            else:
                out = flat_t[torch.from_numpy(np.ascontiguousarray(idx)).to(p.device)].cpu().numpy()
            if self.op == "conj":
                out = np.conj(out)
            return np.asarray(out)
        else:
            # flat window of host memory covering every element this view can reach
            import ctypes
            lo = self.offset + builtins_sum(min(0, (n - 1) * s) for n, s in zip(self.size, self.strides))
            hi = self.offset + builtins_sum(max(0, (n - 1) * s) for n, s in zip(self.size, self.strides))
            isz = self.dtype.itemsize
            raw = (ctypes.c_char * ((hi - lo + 1) * isz)).from_address(self._base + lo * isz)
            flat = np.frombuffer(raw, dtype=self.dtype)
            base_shift = -lo
        if not self.size:
            out = flat[base_shift + self.offset].copy()
        else:
            idx = np.full(self.size, base_shift + self.offset, dtype=np.int64)
            for d, (n, s) in enumerate(zip(self.size, self.strides)):
                shp = [1] * self.ndim
                shp[d] = n
                idx = idx + (np.arange(n, dtype=np.int64) * s).reshape(shp)
            out = flat[idx]
        if self.op == "conj":
            out = np.conj(out)
        return np.asarray(out)

    # operators building lazy Broadcasted trees live in broadcast.py (installed at import)


def isstrided(a) -> bool:
    return isinstance(a, StridedView)


def sreshape(a: StridedView, newsize):
    return a.sreshape(newsize)


def sview(a: StridedView, *idx):
    return a.sview(*idx)
