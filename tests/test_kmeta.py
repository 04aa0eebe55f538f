"""Kernarg layout from the code object's metadata (csrc/smr_kmeta.cpp) -- CPU tests on the BUILT library's own code objects.

Direct dispatch (csrc/smr_seq.cpp) fills the hidden kernel arguments (block counts, group sizes, grid dims, dynamic LDS size) at the
offsets the compiler recorded in the code object's NT_AMDGPU_METADATA note instead of hard-coding the code-object-v5 layout (VERDICT r4,
next-round item 4a).  Here the parser runs over every gfx950 code object embedded in libstrided_hip.so: the fat binary section is cut
into its compressed bundles, each is unbundled with clang-offload-bundler, and every kernel's layout is checked against

  * llvm-readelf's own rendering of the same note (an independent parser), for a sample of kernels, and
  * the code-object-v5 rule the previous round hard-coded (hidden block at the next multiple of 8 behind the explicit arguments: block
    counts +0, group sizes +12, grid dims +64, dynamic LDS +120) -- the rule the fallback path still uses,

and that no kernel of the library asks for a hidden argument only the HIP runtime could supply.
"""
import ctypes as C
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "strided.jl_amd", "libstrided_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("kernarg_size explicit_end nargs_explicit bc_x bc_y bc_z gs_x gs_y gs_z rem_x rem_y rem_z go_x go_y go_z grid_dims dynamic_lds "
          "needs_runtime private_size group_static").split()


def _tools():
    return all(os.access(os.path.join(LLVM, t), os.X_OK) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))


pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and _tools()), reason="needs the built library and the LLVM binary tools")


@pytest.fixture(scope="module")
def code_objects():
    """[(path, bytes)] of every gfx950 code object inside the library"""
    tmp = tempfile.mkdtemp(prefix="smr_kmeta_")
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, LIB, os.path.join(tmp, "copy.so")])
    blob = open(fat, "rb").read()
    out = []
    starts = [m.start() for m in re.finditer(b"CCOB", blob)]
    plain = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
    chunks = []
    for p in starts:   # compressed bundle, header v2: 32-bit sizes; v3: 64-bit sizes
        ver = struct.unpack_from("<H", blob, p + 4)[0]
        size = struct.unpack_from("<Q", blob, p + 8)[0] if ver >= 3 else struct.unpack_from("<I", blob, p + 8)[0]
        chunks.append(blob[p:p + size])
    if not starts:
        for i, p in enumerate(plain):
            chunks.append(blob[p:plain[i + 1] if i + 1 < len(plain) else len(blob)])
    for i, ch in enumerate(chunks):
        b, co = os.path.join(tmp, "b%d" % i), os.path.join(tmp, "co%d.elf" % i)
        open(b, "wb").write(ch)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + b, "--output=" + co], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        data = open(co, "rb").read()
        if data[:4] == b"\x7fELF":
            out.append((co, data))
    assert len(out) >= 30, "expected one code object per kernel translation unit, found %d" % len(out)
    return out


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(LIB)
    L.smr_debug_kernarg_layout.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_int, C.POINTER(C.c_int32), C.c_char_p, C.c_size_t]
    L.smr_debug_kernarg_layout.restype = C.c_int
    return L


def layouts(lib, data):
    buf = C.create_string_buffer(data, len(data))
    n = lib.smr_debug_kernarg_layout(buf, len(data), None, -1, None, None, 0)
    assert n >= 0
    res = {}
    for i in range(n):
        o = (C.c_int32 * 20)()
        name = C.create_string_buffer(1024)
        assert lib.smr_debug_kernarg_layout(buf, len(data), None, i, o, name, 1024) == n
        res[name.value.decode()] = dict(zip(FIELDS, list(o)))
    return res


def test_every_kernel_of_the_library_has_the_v5_layout_and_needs_nothing_from_the_runtime(lib, code_objects):
    total = with_hidden = 0
    for path, data in code_objects:
        for sym, L in layouts(lib, data).items():
            total += 1
            assert sym.endswith(".kd")
            assert L["needs_runtime"] == 0, sym + " declares a hidden argument only HIP can supply"
            hid = (L["explicit_end"] + 7) & ~7
            # a kernel declares only the hidden arguments it reads (one that never asks for blockDim has no hidden block at all);
            # what IS declared sits where code-object-v5 puts it
            rule = {"bc_x": hid, "bc_y": hid + 4, "bc_z": hid + 8, "gs_x": hid + 12, "gs_y": hid + 14, "gs_z": hid + 16,
                    "rem_x": hid + 18, "rem_y": hid + 20, "rem_z": hid + 22, "go_x": hid + 40, "go_y": hid + 48, "go_z": hid + 56,
                    "grid_dims": hid + 64, "dynamic_lds": hid + 120}
            for f, off in rule.items():
                assert L[f] in (-1, off), (sym, f, L[f], off)
                if L[f] >= 0:
                    assert L[f] < L["kernarg_size"], sym
                    with_hidden += 1
            assert L["kernarg_size"] >= L["explicit_end"]
    assert with_hidden > 1000
    assert total > 500, "the library holds hundreds of kernel instantiations, parsed %d" % total


def test_against_llvm_readelf(lib, code_objects):
    """an independent parser of the same note: offsets of every hidden_* / explicit argument, a sample of code objects"""
    for path, data in code_objects[:: max(1, len(code_objects) // 6)]:
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True).stdout
        mine = layouts(lib, data)
        kernels = txt.split("  - .agpr_count:")[1:]
        assert len(kernels) == len(mine)
        for blk in kernels[:40]:
            sym = re.search(r"\.symbol:\s+(\S+)", blk).group(1)
            L = mine[sym]
            assert int(re.search(r"\.kernarg_segment_size:\s+(\d+)", blk).group(1)) == L["kernarg_size"]
            assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)) == L["private_size"]
            args = re.findall(r"- (?:\.\w+:\s+\S+\s+)*?\.offset:\s+(\d+)\s+\.size:\s+(\d+)\s+\.value_kind:\s+(\w+)", blk)
            if not args:  # a kernel without arguments (the window fence)
                assert L["explicit_end"] == 0 and L["nargs_explicit"] == 0, sym
                continue
            want = {"hidden_block_count_x": "bc_x", "hidden_group_size_x": "gs_x", "hidden_grid_dims": "grid_dims",
                    "hidden_dynamic_lds_size": "dynamic_lds", "hidden_remainder_z": "rem_z", "hidden_global_offset_y": "go_y"}
            seen = {k: int(off) for off, _, k in args}
            for k, f in want.items():
                assert L[f] == seen.get(k, -1), (sym, k)
            ends = [int(off) + int(sz) for off, sz, k in args if not k.startswith("hidden_")]
            assert L["explicit_end"] == (max(ends) if ends else 0) and L["nargs_explicit"] == len(ends), sym


def test_malformed_images_are_refused_not_crashed_on(lib, code_objects):
    _, data = code_objects[0]
    for bad in (b"", b"\x7fELF", data[:200], data[:1000], b"\x00" * 4096, bytes(reversed(data[:8192]))):
        buf = C.create_string_buffer(bad, max(1, len(bad)))
        assert lib.smr_debug_kernarg_layout(buf, len(bad), None, -1, None, None, 0) < 0
    # a corrupted note: flip bytes inside the MessagePack document -- any outcome but a crash / hang
    for pos in range(0x400, min(len(data), 0x4000), 0x155):
        m = bytearray(data)
        m[pos] ^= 0xFF
        buf = C.create_string_buffer(bytes(m), len(m))
        lib.smr_debug_kernarg_layout(buf, len(m), None, -1, None, None, 0)


def test_parser_survives_damaged_images_under_asan(code_objects, tmp_path):
    """tools/kmeta_fuzz.cpp: real code objects damaged at random (bytes inside the metadata note, the ELF header, anywhere; images cut
    short) and parsed from an exact-size heap copy with AddressSanitizer + UBSan -- no out-of-bounds read, no undefined behaviour."""
    exe = str(tmp_path / "kmeta_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "strided.jl_amd", "csrc"), os.path.join(ROOT, "tools", "kmeta_fuzz.cpp"),
                        os.path.join(ROOT, "strided.jl_amd", "csrc", "smr_kmeta.cpp"), "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here: " + r.stderr[-200:])
    for seed, (path, _) in enumerate(code_objects[::13][:3]):
        r = subprocess.run([exe, path, "600", str(seed + 1)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "no memory error" in r.stdout, (r.stdout[-300:], r.stderr[-1500:])
        m = re.search(r"(\d+) layouts parsed, (\d+) refused", r.stdout)
        assert int(m.group(1)) > 300 and int(m.group(2)) > 100     # both outcomes occur: the damage is neither harmless nor always fatal
