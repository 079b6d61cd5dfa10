// Host cost of hipModuleLaunchKernel against the size of the argument block (the generated chain kernels take a 3.6 KB
// ChainProgram by value; would a block that holds only what they read launch faster?).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_cost profiles/launch_cost.cpp -lhiprtc && /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));     \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main()
{
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (size_t bytes : { (size_t)64, (size_t)416, (size_t)1024, (size_t)2080, (size_t)3600, (size_t)4096 }) {
        std::string src = "struct A { unsigned int w[" + std::to_string(bytes / 4) + "]; };\n"
                          "extern \"C\" __global__ void k(const A a, unsigned int *out) { if (a.w[0] == 12345u) out[0] = a.w[" +
                          std::to_string(bytes / 4 - 1) + "]; }\n";
        hiprtcProgram prog;
        hiprtcCreateProgram(&prog, src.c_str(), "k.hip", 0, nullptr, nullptr);
        const char *opts[] = { "--gpu-architecture=gfx950", "-O3" };
        if (hiprtcCompileProgram(prog, 2, opts) != HIPRTC_SUCCESS) return 2;
        size_t n = 0;
        hiprtcGetCodeSize(prog, &n);
        std::vector<char> code(n);
        hiprtcGetCode(prog, code.data());
        hipModule_t m;
        hipFunction_t f;
        CK(hipModuleLoadData(&m, code.data()));
        CK(hipModuleGetFunction(&f, m, "k"));
        unsigned int *out = nullptr;
        CK(hipMalloc((void **)&out, 4));
        std::vector<char> args(bytes + 8, 0);
        *(unsigned int **)(args.data() + bytes) = out;
        size_t size = args.size();
        void *extra[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, args.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
        for (int rep = 0; rep < 2; ++rep) {
            const int N = 2000;
            CK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) CK(hipModuleLaunchKernel(f, 64, 1, 1, 256, 1, 1, 0, s, nullptr, extra));
            const auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s));
            const auto t2 = std::chrono::steady_clock::now();
            if (rep)
                std::printf("args %4zu B: host %.2f us per launch (enqueue only), %.2f us per launch incl. drain\n", bytes,
                            std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
                            std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
        }
        CK(hipModuleUnload(m));
        CK(hipFree(out));
    }
    return 0;
}
