// smr_kmeta.h -- kernarg layout of a kernel, read from the code object's own metadata.
//
// Direct dispatch (smr_seq.cpp) writes the kernel-argument block itself: the explicit arguments followed by the hidden ("implicit")
// arguments the compiler reads blockDim / gridDim / the dynamic LDS size from.  Where those hidden fields live is recorded by the
// compiler in the code object: ELF note NT_AMDGPU_METADATA (owner "AMDGPU", type 32), a MessagePack document whose
// amdhsa.kernels[].args[] entries carry .offset / .size / .value_kind.  This file parses exactly that (ELF64 little endian, the
// MessagePack subset the AMDGPU back end emits) -- host-only code, no device needed, unit-tested on the built library's own code
// objects (tests/test_kmeta.py).
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>

namespace smr {

struct KernargLayout {
    int32_t kernarg_size = 0;    // .kernarg_segment_size
    int32_t explicit_end = 0;    // end of the last explicit (non-hidden) argument
    int32_t nargs_explicit = 0;
    int32_t block_count[3] = {-1, -1, -1};   // byte offsets of the hidden fields, -1 = the kernel does not declare the field
    int32_t group_size[3] = {-1, -1, -1};
    int32_t remainder[3] = {-1, -1, -1};
    int32_t global_offset[3] = {-1, -1, -1};
    int32_t grid_dims = -1;
    int32_t dynamic_lds = -1;
    // hidden arguments only the HIP runtime can supply (printf / hostcall buffers, device heap, default queue, completion action,
    // multigrid sync, queue pointer): a kernel that declares one is never dispatched directly
    int32_t needs_runtime = 0;
    int32_t private_size = 0, group_static = 0;
};

// Parses every kernel of the code object at [elf, elf + bytes).  Keys of `out`: the kernel descriptor symbol ("<mangled name>.kd").
// Returns false (and says why) when the image is not an AMDGPU code object with metadata.
bool kmeta_parse(const void* elf, size_t bytes, std::map<std::string, KernargLayout>& out, std::string& why);

// Fills the hidden arguments of `block` (a kernarg block of layout.kernarg_size bytes whose explicit part is already in place) for a
// 1-D launch of `grid` workgroups of `block_size` lanes with `dyn_lds` bytes of dynamic LDS.
void kmeta_fill_hidden(const KernargLayout& layout, unsigned char* block, uint32_t grid, uint32_t block_size, uint32_t dyn_lds);

// The layout code-object-v5 prescribes for a kernel whose explicit arguments end at `explicit_end` (hidden block at the next multiple
// of 8: block counts +0, group sizes +12, remainders +18, global offsets +40, grid dims +64, dynamic LDS size +120).  Used when the
// metadata cannot be read (and then only after the self-test kernel confirmed it, smr_seq.cpp).
KernargLayout kmeta_v5_default(size_t explicit_end, size_t kernarg_size);

}  // namespace smr
