// Memory-floor experiments for the chain kernel's access pattern (6 planes in, 3 out; 4096x4096 f32).
// Not part of the product: a standalone HIP program that answers "what does the HBM system give this
// pattern" under different plane placements, grid shapes and per-thread work.
//   hipcc --offload-arch=gfx950 -O3 profiles/membench.hip -o gpurun_out/membench && gpurun_out/membench
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

struct Args {
    const float4 *a[3];
    const float4 *b[3];
    float4 *o[3];
    uint32_t n4;  // float4 per plane
};

// U float4 per lane, block-contiguous: block handles U*256 consecutive float4 of plane blockIdx.y
template <int U>
__global__ __launch_bounds__(256) void add_y(Args p)
{
    const int ch = blockIdx.y;
    const float4 *__restrict__ a = p.a[ch];
    const float4 *__restrict__ b = p.b[ch];
    float4 *__restrict__ o = p.o[ch];
    const uint32_t base = blockIdx.x * (U * 256u) + threadIdx.x;
    float4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = a[base + u * 256u];
#pragma unroll
    for (int u = 0; u < U; ++u) y[u] = b[base + u * 256u];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float4 r = { x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w };
        o[base + u * 256u] = r;
    }
}

// U float4 per lane, the U chunks of a block spread over the plane (chunk u at u * n4 / U + block * 256)
template <int U>
__global__ __launch_bounds__(256) void add_spread(Args p)
{
    const int ch = blockIdx.y;
    const float4 *__restrict__ a = p.a[ch];
    const float4 *__restrict__ b = p.b[ch];
    float4 *__restrict__ o = p.o[ch];
    const uint32_t part = p.n4 / U;
    const uint32_t base = blockIdx.x * 256u + threadIdx.x;
    float4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = a[base + u * part];
#pragma unroll
    for (int u = 0; u < U; ++u) y[u] = b[base + u * part];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float4 r = { x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w };
        o[base + u * part] = r;
    }
}

// same, but one block does all three channels of its pixel range (9 streams per block)
template <int U>
__global__ __launch_bounds__(256) void add_3ch(Args p)
{
    const uint32_t base = blockIdx.x * (U * 256u) + threadIdx.x;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = p.a[ch][base + u * 256u];
#pragma unroll
        for (int u = 0; u < U; ++u) y[u] = p.b[ch][base + u * 256u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 r = { x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w };
            p.o[ch][base + u * 256u] = r;
        }
    }
}

// grid-stride persistent variant
template <int U>
__global__ __launch_bounds__(256) void add_persist(Args p, uint32_t tiles)
{
    for (uint32_t t = blockIdx.x; t < tiles * 3; t += gridDim.x) {
        const int ch = t / tiles;
        const uint32_t base = (t - ch * tiles) * (U * 256u) + threadIdx.x;
        float4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = p.a[ch][base + u * 256u];
#pragma unroll
        for (int u = 0; u < U; ++u) y[u] = p.b[ch][base + u * 256u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 r = { x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w };
            p.o[ch][base + u * 256u] = r;
        }
    }
}

__global__ __launch_bounds__(256) void copy4(const float4 *__restrict__ a, float4 *__restrict__ o)
{
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x;
    float4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = a[base + u * 256u];
#pragma unroll
    for (int u = 0; u < 4; ++u) o[base + u * 256u] = x[u];
}

__global__ __launch_bounds__(256) void read4(const float4 *__restrict__ a, float *__restrict__ o)
{
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x;
    float4 x[4];
    float s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = a[base + u * 256u];
#pragma unroll
    for (int u = 0; u < 4; ++u) s += x[u].x + x[u].y + x[u].z + x[u].w;
    if (s == 12345.678f) o[0] = s;
}

__global__ __launch_bounds__(256) void write4(float4 *__restrict__ o)
{
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x;
    const float4 v = { 1.f, 2.f, 3.f, 4.f };
#pragma unroll
    for (int u = 0; u < 4; ++u) o[base + u * 256u] = v;
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
    return ms * 1e-3 / reps;
}

int main()
{
    const uint32_t S = 4096;
    const size_t plane = (size_t)S * S * 4;
    const uint32_t n4 = S * S / 4;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const size_t slack = 64u << 20;
    char *buf = nullptr;
    CK(hipMalloc((void **)&buf, 9 * plane + slack));
    CK(hipMemsetAsync(buf, 0, 9 * plane + slack, s));
    const int reps = 50;
    const double bytes9 = 9.0 * plane;

    {
        double t = timed(s, reps, [&] { copy4<<<n4 / 1024, 256, 0, s>>>((const float4 *)buf, (float4 *)(buf + plane)); });
        std::printf("copy 1 plane (fits MALL)      %7.1f us  %6.2f TB/s\n", t * 1e6, 2.0 * plane / t / 1e12);
        t = timed(s, reps, [&] { copy4<<<4 * n4 / 1024, 256, 0, s>>>((const float4 *)buf, (float4 *)(buf + 4 * plane)); });
        std::printf("copy 4 planes (512 MB moved)  %7.1f us  %6.2f TB/s\n", t * 1e6, 8.0 * plane / t / 1e12);
        t = timed(s, reps, [&] { read4<<<8 * n4 / 1024, 256, 0, s>>>((const float4 *)buf, (float *)(buf + 8 * plane)); });
        std::printf("read 8 planes                 %7.1f us  %6.2f TB/s\n", t * 1e6, 8.0 * plane / t / 1e12);
        t = timed(s, reps, [&] { write4<<<8 * n4 / 1024, 256, 0, s>>>((float4 *)buf); });
        std::printf("write 8 planes                %7.1f us  %6.2f TB/s\n", t * 1e6, 8.0 * plane / t / 1e12);
    }

    const size_t skews[] = { 0, 256, 1024, 4096, 4096 + 256, 65536 + 4096 + 256, (1u << 20) + 65536 + 4096 + 256, (2u << 20) + 4096 };
    for (size_t skew : skews) {
        Args p;
        // layout: a0 a1 a2 b0 b1 b2 o0 o1 o2, each displaced by i*skew
        for (int i = 0; i < 3; ++i) {
            p.a[i] = (const float4 *)(buf + (size_t)(i)*(plane + skew));
            p.b[i] = (const float4 *)(buf + (size_t)(3 + i) * (plane + skew));
            p.o[i] = (float4 *)(buf + (size_t)(6 + i) * (plane + skew));
        }
        p.n4 = n4;
        double t1 = timed(s, reps, [&] { add_y<1><<<dim3(n4 / 256, 3), 256, 0, s>>>(p); });
        double t2 = timed(s, reps, [&] { add_y<2><<<dim3(n4 / 512, 3), 256, 0, s>>>(p); });
        double t4 = timed(s, reps, [&] { add_y<4><<<dim3(n4 / 1024, 3), 256, 0, s>>>(p); });
        double t8 = timed(s, reps, [&] { add_y<8><<<dim3(n4 / 2048, 3), 256, 0, s>>>(p); });
        double tsp = timed(s, reps, [&] { add_spread<4><<<dim3(n4 / 1024, 3), 256, 0, s>>>(p); });
        double t3c = timed(s, reps, [&] { add_3ch<2><<<n4 / 512, 256, 0, s>>>(p); });
        double t3c4 = timed(s, reps, [&] { add_3ch<4><<<n4 / 1024, 256, 0, s>>>(p); });
        double tp = timed(s, reps, [&] { add_persist<4><<<256 * 8, 256, 0, s>>>(p, n4 / 1024); });
        double tp2 = timed(s, reps, [&] { add_persist<4><<<256 * 4, 256, 0, s>>>(p, n4 / 1024); });
        std::printf("skew %8zu: y/U1 %6.1f  y/U2 %6.1f  y/U4 %6.1f  spread/U4 %6.1f  y/U8 %6.1f  3ch/U2 %6.1f  3ch/U4 %6.1f  persist8 %6.1f  persist4 %6.1f us  (best %.2f TB/s)\n",
                    skew, t1 * 1e6, t2 * 1e6, t4 * 1e6, tsp * 1e6, t8 * 1e6, t3c * 1e6, t3c4 * 1e6, tp * 1e6, tp2 * 1e6,
                    bytes9 / std::min({ t1, t2, t4, t8, t3c, t3c4, tp, tp2 }) / 1e12);
    }
    CK(hipFree(buf));
    return 0;
}
