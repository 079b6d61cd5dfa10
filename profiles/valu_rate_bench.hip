// What do the band waves' packed multiplies and adds cost on gfx950?  A standalone HIP program (not part of the product).
// One wave per workgroup; every lane keeps C accumulators and runs `iters` rounds of   acc[c] = acc[c] + p * w   as a separate
// multiply and add (the library is built with -ffp-contract=off: the reference rounds twice), packed (v_pk_mul_f32 +
// v_pk_add_f32 on float2) or plain, with the broadcast of one weight out of a register pair (op_sel) as resize_poly_kernel does
// it.  Prints shader clocks (s_memtime) per vector instruction of one wave, alone on its SIMD and with 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/valu_rate_bench.hip -o gpurun_out/valu_rate_bench && gpurun_out/valu_rate_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: packed mul + add, weight splat by op_sel; 1: packed mul + add, weight already a splat; 2: plain v_mul + v_add on
// two floats (4 instructions per mad pair); 3: packed fma (what -ffp-contract=fast would give)
template <int C, int MODE>
__global__ __launch_bounds__(256) void valu(float *out, unsigned long long *clocks, int iters, float seed)
{
    f2 acc[C];
    f2 p[8];
    f2 w[4];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = f2{ seed * c, seed + c };
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] = f2{ seed + u + threadIdx.x, seed - u };
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = f2{ 1.0f + seed * u, 1.0f - seed * u };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                asm volatile("" : "+v"(w[u / 2]));
                const f2 wp = w[u / 2];
                if constexpr (MODE == 0) {
                    acc[c] += p[u] * ((u & 1) ? __builtin_shufflevector(wp, wp, 1, 1) : __builtin_shufflevector(wp, wp, 0, 0));
                } else if constexpr (MODE == 1) {
                    acc[c] += p[u] * wp;
                } else if constexpr (MODE == 2) {
                    acc[c].x += p[u].x * wp.x;
                    acc[c].y += p[u].y * wp.x;
                } else {
                    acc[c] = __builtin_elementwise_fma(p[u], wp, acc[c]);
                }
            }
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(p[u]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    f2 s = acc[0];
#pragma unroll
    for (int c = 1; c < C; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

template <int C, int MODE>
static void run(const char *what, float *out, unsigned long long *clocks, int vinst_per_mad)
{
    const int iters = 2000;
    for (int threads : { 64, 128, 256, 512 })
        for (int wgs : { 1, 256, 1024 }) {
            if (threads > 256) continue;
            valu<C, MODE><<<wgs, threads>>>(out, clocks, iters, 0.001f);
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            valu<C, MODE><<<wgs, threads>>>(out, clocks, iters, 0.001f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long sum = 0;
            for (int i = 0; i < wgs; ++i) sum += clocks[i];
            const double per = (double)sum / wgs / ((double)iters * 8 * C * vinst_per_mad);
            std::printf("%-44s C=%d  %4d wgs x %d waves: %6.2f clocks per vector instruction per wave, kernel %7.1f us (%.2f ns per instruction per wave)\n", what, C, wgs,
                        threads / 64, per, ms * 1e3, ms * 1e6 / ((double)iters * 8 * C * vinst_per_mad));
        }
}

int main()
{
    float *out;
    unsigned long long *clocks;
    CK(hipMalloc(&out, 1 << 22));
    CK(hipHostMalloc(&clocks, 1024 * 8));
    run<6, 0>("pk_mul + pk_add, op_sel splat", out, clocks, 2);
    run<12, 0>("pk_mul + pk_add, op_sel splat", out, clocks, 2);
    run<6, 1>("pk_mul + pk_add, no splat", out, clocks, 2);
    run<6, 2>("v_mul + v_add (two floats)", out, clocks, 4);
    run<6, 3>("pk_fma", out, clocks, 1);
    run<2, 0>("pk_mul + pk_add, op_sel splat", out, clocks, 2);
    return 0;
}
