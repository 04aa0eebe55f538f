"""Shared helpers of the parity tests: run one funnel call through the CPU oracle (host views)
and through the HIP library (device views with the identical layout)."""
import numpy as np

import oraclelib
import strided_jl_amd as S


def fview(arr):
    """Host StridedView over a fresh column-major copy of `arr` (Julia's layout)."""
    return S.StridedView(np.asfortranarray(arr).copy(order="F"))


def window(view):
    """(lo, hi) element range of the parent memory the view can touch."""
    lo = view.offset + sum(min(0, (n - 1) * s) for n, s in zip(view.size, view.strides))
    hi = view.offset + sum(max(0, (n - 1) * s) for n, s in zip(view.size, view.strides))
    return lo, hi


def host_flat(view):
    """1-d numpy array aliasing the WHOLE root allocation of a host view + the element index of
    the view's base inside it."""
    root = view.parent
    while isinstance(root.base, np.ndarray):
        root = root.base
    import ctypes
    n = root.size
    raw = (ctypes.c_char * (n * root.itemsize)).from_address(root.ctypes.data)
    flat = np.frombuffer(raw, dtype=root.dtype)
    shift = (view._base - root.ctypes.data) // root.itemsize
    return flat, shift


def to_device(view, cache=None):
    """Device StridedView with the same (size, strides, offset, op) over a device copy of the
    host view's root allocation.  `cache` maps id(root) -> torch tensor so that views sharing a
    parent on the host share one on the device too (aliasing is preserved)."""
    import torch
    flat, shift = host_flat(view)
    key = flat.ctypes.data
    if cache is not None and key in cache:
        t = cache[key]
    else:
        t = torch.from_numpy(flat.copy()).cuda()
        if cache is not None:
            cache[key] = t
    return S.StridedView(t, view.size, view.strides, view.offset + shift, view.op)


def run_oracle(f, op, initop, dims, arrays, nthreads=1):
    """arrays: host StridedViews (arrays[0] = destination, modified in place)."""
    p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
    oraclelib.mapreduce(p, nthreads)
    return arrays[0].toarray()


def run_device(f, op, initop, dims, arrays):
    """Same call on device copies; returns the destination as a host array."""
    import torch
    cache = {}
    dev = tuple(to_device(a, cache) for a in arrays)
    S._mapreduce_fuse_(f, op, initop, dims, dev)
    torch.cuda.synchronize()
    return dev[0].toarray()


def rand(rng, shape, dtype):
    dtype = np.dtype(dtype)
    if np.issubdtype(dtype, np.complexfloating):
        r = rng.random(shape) + 1j * rng.random(shape)
    else:
        r = rng.random(shape)
    return np.asfortranarray(r.astype(dtype))


def randn(rng, shape, dtype):
    dtype = np.dtype(dtype)
    if np.issubdtype(dtype, np.complexfloating):
        r = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    else:
        r = rng.standard_normal(shape)
    return np.asfortranarray(r.astype(dtype))


FLOATS = [np.float32, np.float64, np.complex64, np.complex128]


def rtol(dtype):
    """Julia's isapprox default: sqrt(eps) of the real type."""
    return float(np.sqrt(np.finfo(np.dtype(dtype)).eps))
