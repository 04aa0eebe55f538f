// smr_jitc -- out-of-process compiler used by libstrided_hip.so (smr_jit.cpp).
//
//   smr_jitc <source file> <code object file>
//
// Compiles one generated translation unit (a kernel family's source, embedded below, plus the
// functor generated from an f-program) for gfx950 with hiprtc and writes the code object.  It is a
// separate process on purpose: the host application (PyTorch, Julia, ...) usually carries its own
// copies of LLVM / comgr, and running hiprtc inside such a process was observed to crash on some
// inputs (symbol interposition between the LLVM copies); a clean process has exactly one.
#include <hip/hiprtc.h>

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

extern "C" const int smr_embed_count;
extern "C" const char* const smr_embed_names[];
extern "C" const char* const smr_embed_texts[];

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: smr_jitc <source> <output>\n");
        return 2;
    }
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) {
        std::fprintf(stderr, "smr_jitc: cannot read %s\n", argv[1]);
        return 2;
    }
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string src = ss.str();
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "smr_jit.hip", smr_embed_count, (const char**)smr_embed_texts,
                            (const char**)smr_embed_names) != HIPRTC_SUCCESS) {
        std::fprintf(stderr, "smr_jitc: hiprtcCreateProgram failed\n");
        return 1;
    }
    // -fwrapv: the integer compute class wraps like Julia's Int64 (no effect on the floating classes)
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fwrapv"};
    const hiprtcResult r = hiprtcCompileProgram(prog, 5, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n > 1) hiprtcGetProgramLog(prog, &log[0]);
        std::fprintf(stderr, "smr_jitc: hiprtcCompileProgram failed (%d):\n%s\n", (int)r, log.c_str());
        return 1;
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    std::string code(n, '\0');
    if (hiprtcGetCode(prog, &code[0]) != HIPRTC_SUCCESS) {
        std::fprintf(stderr, "smr_jitc: hiprtcGetCode failed\n");
        return 1;
    }
    std::ofstream out(argv[2], std::ios::binary);
    out.write(code.data(), (std::streamsize)code.size());
    out.close();
    if (!out) {
        std::fprintf(stderr, "smr_jitc: cannot write %s\n", argv[2]);
        return 2;
    }
    hiprtcDestroyProgram(&prog);
    return 0;
}
