// What a program-specialised (straight-line) chain kernel can reach on the headline workload, before
// building the specialiser: the 32-node BASELINE graph = 16 records "acc = 1 - (acc op B)" with op
// alternating +, *, on 3 channels of 4096x4096 f32 (6 planes in, 3 out).  Variants: float4 per lane (U),
// workgroup size, channel mapping (blockIdx.y vs interleaved in blockIdx.x).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/straightline.hip -o gpurun_out/straightline
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

struct Args {
    const f4 *a[3];
    const f4 *b[3];
    f4 *o[3];
    uint32_t n4;
    float c;
};

template <int NREC>
static __device__ __forceinline__ f4 program(f4 acc, f4 y, float c)
{
#pragma unroll
    for (int i = 0; i < NREC; ++i) {
        if (i & 1)
            acc = c - acc * y;
        else
            acc = c - (acc + y);
    }
    return acc;
}

template <int U, int NREC, int WG>
__global__ __launch_bounds__(WG) void sl_y(Args p)
{
    const int ch = blockIdx.y;
    const f4 *__restrict__ a = p.a[ch];
    const f4 *__restrict__ b = p.b[ch];
    f4 *__restrict__ o = p.o[ch];
    const uint32_t base = blockIdx.x * (U * WG) + threadIdx.x;
    f4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = a[base + u * WG];
#pragma unroll
    for (int u = 0; u < U; ++u) y[u] = b[base + u * WG];
#pragma unroll
    for (int u = 0; u < U; ++u) o[base + u * WG] = program<NREC>(x[u], y[u], p.c);
}

// channel interleaved in blockIdx.x: consecutive workgroups work on the same pixels of R, G, B
template <int U, int NREC, int WG>
__global__ __launch_bounds__(WG) void sl_x3(Args p)
{
    const uint32_t ch = blockIdx.x % 3u, t = blockIdx.x / 3u;
    const f4 *__restrict__ a = p.a[ch];
    const f4 *__restrict__ b = p.b[ch];
    f4 *__restrict__ o = p.o[ch];
    const uint32_t base = t * (U * WG) + threadIdx.x;
    f4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = a[base + u * WG];
#pragma unroll
    for (int u = 0; u < U; ++u) y[u] = b[base + u * WG];
#pragma unroll
    for (int u = 0; u < U; ++u) o[base + u * WG] = program<NREC>(x[u], y[u], p.c);
}

// grid-stride with a software pipeline: loads of trip i + 1 are issued before trip i is computed
template <int NREC>
__global__ __launch_bounds__(256) void sl_pipe(Args p, uint32_t tiles)
{
    uint32_t t = blockIdx.x;
    if (t >= tiles * 3) return;
    int ch = t / tiles;
    uint32_t base = (t - ch * tiles) * 256u + threadIdx.x;
    f4 x = p.a[ch][base], y = p.b[ch][base];
    for (;;) {
        const uint32_t tn = t + gridDim.x;
        f4 xn = x, yn = y;
        int chn = ch;
        uint32_t basen = base;
        if (tn < tiles * 3) {
            chn = tn / tiles;
            basen = (tn - chn * tiles) * 256u + threadIdx.x;
            xn = p.a[chn][basen];
            yn = p.b[chn][basen];
        }
        p.o[ch][base] = program<NREC>(x, y, p.c);
        if (tn >= tiles * 3) break;
        t = tn;
        ch = chn;
        base = basen;
        x = xn;
        y = yn;
    }
}

template <class F>
static double timed(hipStream_t s, int reps, F f)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms * 1e-3 / reps * 1e6;
}

int main()
{
    const uint32_t S = 4096;
    const size_t plane = (size_t)S * S * 4;
    const uint32_t n4 = S * S / 4;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    char *buf = nullptr;
    CK(hipMalloc((void **)&buf, 9 * plane));
    CK(hipMemsetAsync(buf, 0, 9 * plane, s));
    Args p;
    for (int i = 0; i < 3; ++i) {
        p.a[i] = (const f4 *)(buf + (size_t)i * plane);
        p.b[i] = (const f4 *)(buf + (size_t)(3 + i) * plane);
        p.o[i] = (f4 *)(buf + (size_t)(6 + i) * plane);
    }
    p.n4 = n4;
    p.c = 1.0f;
    const int reps = 100;
    const double bytes = 9.0 * plane;
    auto show = [&](const char *name, double us) { std::printf("%-34s %7.2f us  %5.2f TB/s  frac %.3f\n", name, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0); };
    for (int round = 0; round < 2; ++round) {
        std::printf("--- round %d ---\n", round);
        show("1 record  y U1 wg256", timed(s, reps, [&] { sl_y<1, 1, 256><<<dim3(n4 / 256, 3), 256, 0, s>>>(p); }));
        show("16 records y U1 wg256", timed(s, reps, [&] { sl_y<1, 16, 256><<<dim3(n4 / 256, 3), 256, 0, s>>>(p); }));
        show("16 records y U2 wg256", timed(s, reps, [&] { sl_y<2, 16, 256><<<dim3(n4 / 512, 3), 256, 0, s>>>(p); }));
        show("16 records y U4 wg256", timed(s, reps, [&] { sl_y<4, 16, 256><<<dim3(n4 / 1024, 3), 256, 0, s>>>(p); }));
        show("16 records y U1 wg512", timed(s, reps, [&] { sl_y<1, 16, 512><<<dim3(n4 / 512, 3), 512, 0, s>>>(p); }));
        show("16 records y U1 wg1024", timed(s, reps, [&] { sl_y<1, 16, 1024><<<dim3(n4 / 1024, 3), 1024, 0, s>>>(p); }));
        show("16 records y U2 wg512", timed(s, reps, [&] { sl_y<2, 16, 512><<<dim3(n4 / 1024, 3), 512, 0, s>>>(p); }));
        show("16 records y U1 wg128", timed(s, reps, [&] { sl_y<1, 16, 128><<<dim3(n4 / 128, 3), 128, 0, s>>>(p); }));
        show("16 records y U1 wg64", timed(s, reps, [&] { sl_y<1, 16, 64><<<dim3(n4 / 64, 3), 64, 0, s>>>(p); }));
        show("16 records x3 U1 wg256", timed(s, reps, [&] { sl_x3<1, 16, 256><<<dim3(3 * n4 / 256), 256, 0, s>>>(p); }));
        show("16 records x3 U2 wg256", timed(s, reps, [&] { sl_x3<2, 16, 256><<<dim3(3 * n4 / 512), 256, 0, s>>>(p); }));
        show("32 records y U1 wg256", timed(s, reps, [&] { sl_y<1, 32, 256><<<dim3(n4 / 256, 3), 256, 0, s>>>(p); }));
        show("64 records y U1 wg256", timed(s, reps, [&] { sl_y<1, 64, 256><<<dim3(n4 / 256, 3), 256, 0, s>>>(p); }));
        show("64 records y U2 wg256", timed(s, reps, [&] { sl_y<2, 64, 256><<<dim3(n4 / 512, 3), 256, 0, s>>>(p); }));
        for (uint32_t g : { 2048u, 4096u, 8192u, 16384u }) {
            char nm[64];
            std::snprintf(nm, sizeof nm, "16 records pipe grid %u", g);
            show(nm, timed(s, reps, [&] { sl_pipe<16><<<g, 256, 0, s>>>(p, n4 / 256); }));
        }
    }
    CK(hipFree(buf));
    return 0;
}
