// smr_kmeta.cpp -- see smr_kmeta.h.  ELF64 + the MessagePack subset of NT_AMDGPU_METADATA; host-only.
#include "smr_kmeta.h"

#include <algorithm>
#include <cstring>
#include <iterator>
#include <vector>

namespace smr {
namespace {

// ---- MessagePack cursor (big-endian payloads; every read is bounds-checked, a malformed document ends the parse) -----------------
struct Cur {
    const unsigned char* p;
    const unsigned char* end;
    bool bad = false;

    bool need(size_t n) {
        if ((size_t)(end - p) < n) bad = true;
        return !bad;
    }
    uint64_t be(int n) {
        if (!need((size_t)n)) return 0;
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) v = (v << 8) | p[i];
        p += n;
        return v;
    }
    enum Kind { NIL, BOOL, INT, STR, ARR, MAP, OTHER, BAD };
    struct Tok {
        Kind kind = BAD;
        int64_t i = 0;          // INT / BOOL value; ARR / MAP: number of entries
        const char* s = nullptr;  // STR
        size_t n = 0;
    };
    // reads ONE token; containers are entered (their entries follow), strings / binaries / extensions are consumed
    Tok next() {
        Tok t;
        if (!need(1)) return t;
        const unsigned char c = *p++;
        auto str = [&](size_t n) {
            if (!need(n)) return;
            t.kind = STR;
            t.s = (const char*)p;
            t.n = n;
            p += n;
        };
        auto skip = [&](size_t n) {
            if (!need(n)) return;
            t.kind = OTHER;
            p += n;
        };
        if (c <= 0x7f) {
            t.kind = INT;
            t.i = c;
        } else if (c <= 0x8f) {
            t.kind = MAP;
            t.i = c & 0x0f;
        } else if (c <= 0x9f) {
            t.kind = ARR;
            t.i = c & 0x0f;
        } else if (c <= 0xbf) {
            str(c & 0x1f);
        } else if (c >= 0xe0) {
            t.kind = INT;
            t.i = (int8_t)c;
        } else
            switch (c) {
                case 0xc0: t.kind = NIL; break;
                case 0xc2: t.kind = BOOL; t.i = 0; break;
                case 0xc3: t.kind = BOOL; t.i = 1; break;
                case 0xc4: skip((size_t)be(1)); break;
                case 0xc5: skip((size_t)be(2)); break;
                case 0xc6: skip((size_t)be(4)); break;
                case 0xc7: { const size_t n = (size_t)be(1); skip(n + 1); break; }
                case 0xc8: { const size_t n = (size_t)be(2); skip(n + 1); break; }
                case 0xc9: { const size_t n = (size_t)be(4); skip(n + 1); break; }
                case 0xca: skip(4); break;
                case 0xcb: skip(8); break;
                case 0xcc: t.kind = INT; t.i = (int64_t)be(1); break;
                case 0xcd: t.kind = INT; t.i = (int64_t)be(2); break;
                case 0xce: t.kind = INT; t.i = (int64_t)be(4); break;
                case 0xcf: t.kind = INT; t.i = (int64_t)be(8); break;
                case 0xd0: t.kind = INT; t.i = (int8_t)be(1); break;
                case 0xd1: t.kind = INT; t.i = (int16_t)be(2); break;
                case 0xd2: t.kind = INT; t.i = (int32_t)be(4); break;
                case 0xd3: t.kind = INT; t.i = (int64_t)be(8); break;
                case 0xd4: skip(2); break;
                case 0xd5: skip(3); break;
                case 0xd6: skip(5); break;
                case 0xd7: skip(9); break;
                case 0xd8: skip(17); break;
                case 0xd9: str((size_t)be(1)); break;
                case 0xda: str((size_t)be(2)); break;
                case 0xdb: str((size_t)be(4)); break;
                case 0xdc: t.kind = ARR; t.i = (int64_t)be(2); break;
                case 0xdd: t.kind = ARR; t.i = (int64_t)be(4); break;
                case 0xde: t.kind = MAP; t.i = (int64_t)be(2); break;
                case 0xdf: t.kind = MAP; t.i = (int64_t)be(4); break;
                default: bad = true; break;  // 0xc1: never used
            }
        if (bad) t.kind = BAD;
        return t;
    }
    // consumes one complete value (a container with everything inside it); iterative: depth is bounded by the document size
    void skip_value() {
        uint64_t pending = 1;
        while (pending && !bad) {
            const Tok t = next();
            --pending;
            if (t.kind == ARR) pending += (uint64_t)t.i;
            else if (t.kind == MAP) pending += 2 * (uint64_t)t.i;
            else if (t.kind == BAD) bad = true;
        }
    }
};

bool is(const Cur::Tok& t, const char* s) { return t.kind == Cur::STR && t.n == std::strlen(s) && std::memcmp(t.s, s, t.n) == 0; }

// one entry of .args
void parse_arg(Cur& c, KernargLayout& L) {
    const Cur::Tok m = c.next();
    if (m.kind != Cur::MAP) {
        c.bad = true;
        return;
    }
    int64_t off = -1, size = 0;
    std::string kind;
    for (int64_t i = 0; i < m.i && !c.bad; ++i) {
        const Cur::Tok k = c.next();
        if (is(k, ".offset")) {
            const Cur::Tok v = c.next();
            if (v.kind == Cur::INT) off = v.i; else c.bad = true;
        } else if (is(k, ".size")) {
            const Cur::Tok v = c.next();
            if (v.kind == Cur::INT) size = v.i; else c.bad = true;
        } else if (is(k, ".value_kind")) {
            const Cur::Tok v = c.next();
            if (v.kind == Cur::STR) kind.assign(v.s, v.n); else c.bad = true;
        } else {
            c.skip_value();
        }
    }
    if (c.bad || off < 0) return;
    const int32_t o = (int32_t)off;
    if (kind.compare(0, 7, "hidden_") != 0) {  // by_value, global_buffer, dynamic_shared_pointer, image, sampler, pipe, queue
        L.explicit_end = std::max<int32_t>(L.explicit_end, (int32_t)(off + size));
        ++L.nargs_explicit;
        return;
    }
    static const char* const xyz[3] = {"_x", "_y", "_z"};
    for (int d = 0; d < 3; ++d) {
        if (kind == std::string("hidden_block_count") + xyz[d]) { L.block_count[d] = o; return; }
        if (kind == std::string("hidden_group_size") + xyz[d]) { L.group_size[d] = o; return; }
        if (kind == std::string("hidden_remainder") + xyz[d]) { L.remainder[d] = o; return; }
        if (kind == std::string("hidden_global_offset") + xyz[d]) { L.global_offset[d] = o; return; }
    }
    if (kind == "hidden_grid_dims") { L.grid_dims = o; return; }
    if (kind == "hidden_dynamic_lds_size") { L.dynamic_lds = o; return; }
    if (kind == "hidden_none" || kind == "hidden_private_base" || kind == "hidden_shared_base") return;  // padding / apertures (gfx9+: from registers)
    // hidden_printf_buffer, hidden_hostcall_buffer, hidden_default_queue, hidden_completion_action, hidden_multigrid_sync_arg,
    // hidden_heap_v1, hidden_queue_ptr, and whatever a later compiler adds: only the runtime can fill these
    L.needs_runtime = 1;
}

void parse_kernel(Cur& c, std::map<std::string, KernargLayout>& out) {
    const Cur::Tok m = c.next();
    if (m.kind != Cur::MAP) {
        c.bad = true;
        return;
    }
    KernargLayout L;
    std::string symbol, name;
    for (int64_t i = 0; i < m.i && !c.bad; ++i) {
        const Cur::Tok k = c.next();
        auto int_into = [&](int32_t& dst) {
            const Cur::Tok v = c.next();
            if (v.kind == Cur::INT) dst = (int32_t)v.i; else c.bad = true;
        };
        if (is(k, ".symbol")) {
            const Cur::Tok v = c.next();
            if (v.kind == Cur::STR) symbol.assign(v.s, v.n); else c.bad = true;
        } else if (is(k, ".name")) {
            const Cur::Tok v = c.next();
            if (v.kind == Cur::STR) name.assign(v.s, v.n); else c.bad = true;
        } else if (is(k, ".kernarg_segment_size")) {
            int_into(L.kernarg_size);
        } else if (is(k, ".private_segment_fixed_size")) {
            int_into(L.private_size);
        } else if (is(k, ".group_segment_fixed_size")) {
            int_into(L.group_static);
        } else if (is(k, ".args")) {
            const Cur::Tok a = c.next();
            if (a.kind != Cur::ARR) {
                c.bad = true;
                break;
            }
            for (int64_t j = 0; j < a.i && !c.bad; ++j) parse_arg(c, L);
        } else {
            c.skip_value();
        }
    }
    if (c.bad) return;
    if (symbol.empty() && !name.empty()) symbol = name + ".kd";
    if (!symbol.empty()) out[symbol] = L;
}

bool parse_metadata(const unsigned char* p, size_t n, std::map<std::string, KernargLayout>& out, std::string& why) {
    Cur c{p, p + n};
    const Cur::Tok top = c.next();
    if (top.kind != Cur::MAP) {
        why = "AMDGPU metadata: the document is not a map";
        return false;
    }
    bool seen = false;
    for (int64_t i = 0; i < top.i && !c.bad; ++i) {
        const Cur::Tok k = c.next();
        if (is(k, "amdhsa.kernels")) {
            const Cur::Tok a = c.next();
            if (a.kind != Cur::ARR) {
                c.bad = true;
                break;
            }
            seen = true;
            for (int64_t j = 0; j < a.i && !c.bad; ++j) parse_kernel(c, out);
        } else {
            c.skip_value();
        }
    }
    if (c.bad) {
        why = "AMDGPU metadata: malformed MessagePack";
        return false;
    }
    if (!seen) why = "AMDGPU metadata: no amdhsa.kernels";
    return seen;
}

// ---- ELF64 (little endian) -----------------------------------------------------------------------------------------------------
struct Ehdr {
    unsigned char ident[16];
    uint16_t type, machine;
    uint32_t version;
    uint64_t entry, phoff, shoff;
    uint32_t flags;
    uint16_t ehsize, phentsize, phnum, shentsize, shnum, shstrndx;
};
struct Shdr {
    uint32_t name, type;
    uint64_t flags, addr, offset, size;
    uint32_t link, info;
    uint64_t addralign, entsize;
};
struct Phdr {
    uint32_t type, flags;
    uint64_t offset, vaddr, paddr, filesz, memsz, align;
};
constexpr uint32_t SHT_NOTE_ = 7, PT_NOTE_ = 4, NT_AMDGPU_METADATA_ = 32;

// walks one note region; true when the AMDGPU metadata note was found (and parsed)
bool walk_notes(const unsigned char* p, size_t n, size_t align, std::map<std::string, KernargLayout>& out, std::string& why, bool& found) {
    if (align < 4) align = 4;
    size_t off = 0;
    while (off + 12 <= n) {
        uint32_t namesz, descsz, type;
        std::memcpy(&namesz, p + off, 4);
        std::memcpy(&descsz, p + off + 4, 4);
        std::memcpy(&type, p + off + 8, 4);
        off += 12;
        const size_t name_pad = (namesz + align - 1) / align * align, desc_pad = ((size_t)descsz + align - 1) / align * align;
        if (off + name_pad > n || off + name_pad + descsz > n) break;
        const char* name = (const char*)p + off;
        const unsigned char* desc = p + off + name_pad;
        if (type == NT_AMDGPU_METADATA_ && namesz >= 6 && std::memcmp(name, "AMDGPU", 6) == 0) {
            found = true;
            return parse_metadata(desc, descsz, out, why);
        }
        off += name_pad + desc_pad;
    }
    return false;
}

}  // namespace

bool kmeta_parse(const void* elf, size_t bytes, std::map<std::string, KernargLayout>& out, std::string& why) {
    const unsigned char* b = (const unsigned char*)elf;
    if (!b || bytes < sizeof(Ehdr) || std::memcmp(b, "\177ELF", 4) != 0 || b[4] != 2 /*ELFCLASS64*/ || b[5] != 1 /*little endian*/) {
        why = "not an ELF64 little-endian image";
        return false;
    }
    Ehdr eh;
    std::memcpy(&eh, b, sizeof eh);
    if (eh.machine != 224 /*EM_AMDGPU*/) {
        why = "not an AMDGPU code object (e_machine != 224)";
        return false;
    }
    bool found = false;
    if (eh.shoff && eh.shentsize == sizeof(Shdr) && eh.shoff + (uint64_t)eh.shnum * sizeof(Shdr) <= bytes)
        for (unsigned i = 0; i < eh.shnum && !found; ++i) {
            Shdr sh;
            std::memcpy(&sh, b + eh.shoff + (size_t)i * sizeof(Shdr), sizeof sh);
            if (sh.type != SHT_NOTE_ || sh.offset > bytes || sh.size > bytes - sh.offset) continue;
            if (walk_notes(b + sh.offset, (size_t)sh.size, (size_t)sh.addralign, out, why, found)) return true;
            if (found) return false;
        }
    if (eh.phoff && eh.phentsize == sizeof(Phdr) && eh.phoff + (uint64_t)eh.phnum * sizeof(Phdr) <= bytes)
        for (unsigned i = 0; i < eh.phnum && !found; ++i) {
            Phdr ph;
            std::memcpy(&ph, b + eh.phoff + (size_t)i * sizeof(Phdr), sizeof ph);
            if (ph.type != PT_NOTE_ || ph.offset > bytes || ph.filesz > bytes - ph.offset) continue;
            if (walk_notes(b + ph.offset, (size_t)ph.filesz, (size_t)ph.align, out, why, found)) return true;
            if (found) return false;
        }
    if (why.empty()) why = "no NT_AMDGPU_METADATA note in the code object";
    return false;
}

void kmeta_fill_hidden(const KernargLayout& L, unsigned char* block, uint32_t grid, uint32_t block_size, uint32_t dyn_lds) {
    auto put = [&](int32_t off, const void* v, size_t n) {
        if (off >= 0 && (size_t)off + n <= (size_t)L.kernarg_size) std::memcpy(block + off, v, n);
    };
    if (L.kernarg_size > L.explicit_end) {  // everything hidden the kernel does not get from us reads as zero
        const size_t from = ((size_t)L.explicit_end + 7) & ~(size_t)7;
        if (from < (size_t)L.kernarg_size) std::memset(block + from, 0, (size_t)L.kernarg_size - from);
    }
    const uint32_t counts[3] = {grid, 1, 1};
    const uint16_t sizes[3] = {(uint16_t)block_size, 1, 1}, zero16 = 0, dims = 1;
    const uint64_t zero64 = 0;
    for (int d = 0; d < 3; ++d) {
        put(L.block_count[d], &counts[d], 4);
        put(L.group_size[d], &sizes[d], 2);
        put(L.remainder[d], &zero16, 2);          // grids are whole multiples of the workgroup size
        put(L.global_offset[d], &zero64, 8);
    }
    put(L.grid_dims, &dims, 2);
    put(L.dynamic_lds, &dyn_lds, 4);
}

KernargLayout kmeta_v5_default(size_t explicit_end, size_t kernarg_size) {
    KernargLayout L;
    L.kernarg_size = (int32_t)kernarg_size;
    L.explicit_end = (int32_t)explicit_end;
    const int32_t hid = (int32_t)((explicit_end + 7) & ~(size_t)7);
    auto fits = [&](int32_t off, int32_t n) { return (size_t)(off + n) <= kernarg_size ? off : -1; };
    for (int d = 0; d < 3; ++d) {
        L.block_count[d] = fits(hid + 4 * d, 4);
        L.group_size[d] = fits(hid + 12 + 2 * d, 2);
        L.remainder[d] = fits(hid + 18 + 2 * d, 2);
        L.global_offset[d] = fits(hid + 40 + 8 * d, 8);
    }
    L.grid_dims = fits(hid + 64, 2);
    L.dynamic_lds = fits(hid + 120, 4);
    return L;
}

}  // namespace smr

// ---- C entry point for the tests (no device needed) ---------------------------------------------------------------------------------
// out[0..19] = kernarg_size, explicit_end, nargs_explicit, block_count[3], group_size[3], remainder[3], global_offset[3], grid_dims,
// dynamic_lds, needs_runtime, private_size, group_static.  `symbol`: "<mangled>.kd", or nullptr with index >= 0 to enumerate (the
// symbol is then copied into name_out).  Returns the number of kernels in the code object, or a negative value on a parse error.
extern "C" int smr_debug_kernarg_layout(const void* elf, size_t bytes, const char* symbol, int index, int32_t* out, char* name_out, size_t name_cap) {
    std::map<std::string, smr::KernargLayout> all;
    std::string why;
    if (!smr::kmeta_parse(elf, bytes, all, why)) return -1;
    const smr::KernargLayout* L = nullptr;
    if (symbol) {
        auto it = all.find(symbol);
        if (it != all.end()) L = &it->second;
    } else if (index >= 0 && (size_t)index < all.size()) {
        auto it = all.begin();
        std::advance(it, index);
        L = &it->second;
        if (name_out && name_cap) {
            std::strncpy(name_out, it->first.c_str(), name_cap - 1);
            name_out[name_cap - 1] = 0;
        }
    }
    if (L && out) {
        int k = 0;
        out[k++] = L->kernarg_size;
        out[k++] = L->explicit_end;
        out[k++] = L->nargs_explicit;
        for (int d = 0; d < 3; ++d) out[k++] = L->block_count[d];
        for (int d = 0; d < 3; ++d) out[k++] = L->group_size[d];
        for (int d = 0; d < 3; ++d) out[k++] = L->remainder[d];
        for (int d = 0; d < 3; ++d) out[k++] = L->global_offset[d];
        out[k++] = L->grid_dims;
        out[k++] = L->dynamic_lds;
        out[k++] = L->needs_runtime;
        out[k++] = L->private_size;
        out[k++] = L->group_static;
    }
    if ((symbol || index >= 0) && !L) return -2;
    return (int)all.size();
}
