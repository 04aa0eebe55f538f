// kmeta_fuzz.cpp -- the code-object metadata parser (csrc/smr_kmeta.cpp) on damaged images, under AddressSanitizer / UBSan.
// The parser reads code objects as the loader keeps them in memory; direct dispatch trusts what it returns.  Here a real code object
// is damaged in every way a seed allows (bytes flipped inside the note / the section table / anywhere, the image cut short, lengths
// raised) and parsed for its first kernels; any out-of-bounds read or undefined behaviour aborts the process.
// Build + run: tests/test_kmeta.py::test_parser_survives_damaged_images_under_asan (g++ -fsanitize=address,undefined).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int smr_debug_kernarg_layout(const void* elf, size_t bytes, const char* symbol, int index, int32_t* out, char* name_out, size_t name_cap);

static uint64_t rng_state;
static uint32_t rnd() {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<unsigned char> img;
    unsigned char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) img.insert(img.end(), buf, buf + n);
    std::fclose(f);
    const int rounds = std::atoi(argv[2]);
    rng_state = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1;
    // where the metadata note lives (found by its owner name) -- most damage goes there
    size_t note = 0;
    for (size_t i = 0; i + 6 < img.size(); ++i)
        if (!std::memcmp(&img[i], "AMDGPU\0", 7)) { note = i; break; }
    long parsed = 0, refused = 0;
    for (int r = 0; r < rounds; ++r) {
        std::vector<unsigned char> m(img);
        const int kind = (int)(rnd() % 6);
        const int hits = 1 + (int)(rnd() % 8);
        for (int h = 0; h < hits; ++h) {
            size_t at;
            if (kind <= 2 && note) at = note + rnd() % (size_t)(m.size() - note < 20000 ? m.size() - note : 20000);  // inside the note
            else if (kind == 3) at = rnd() % 64;                                                                      // ELF header
            else at = rnd() % m.size();
            m[at] = kind == 2 ? (unsigned char)0xff : (unsigned char)rnd();
        }
        size_t len = m.size();
        if (kind == 4) len = rnd() % m.size();   // cut short
        // an exact-size heap copy: reads past `len` are caught
        unsigned char* exact = (unsigned char*)std::malloc(len ? len : 1);
        std::memcpy(exact, m.data(), len);
        for (int k = 0; k < 3; ++k) {
            int32_t out[20];
            char name[256];
            const int rc = smr_debug_kernarg_layout(exact, len, nullptr, k, out, name, sizeof name);
            if (rc > 0) ++parsed;
            else ++refused;
        }
        std::free(exact);
    }
    std::printf("%d damaged images: %ld layouts parsed, %ld refused, no memory error\n", rounds, parsed, refused);
    return 0;
}
