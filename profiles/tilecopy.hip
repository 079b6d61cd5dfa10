// What does the HBM system give a 1-in-1-out stream over 3 planes of 4096 x 4096 f32 (config #2's resident traffic:
// 201 MB read + 201 MB written) under different workgroup -> data mappings?  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 profiles/tilecopy.hip -o /tmp/tilecopy && /tmp/tilecopy
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
    f4 *o[3];
    uint32_t w4, h;  // float4 per row, rows
};

static __device__ __forceinline__ f4 work(f4 x) { return (x + 0.25f) * x - 0.25f; }

// flat: block handles U * 256 consecutive float4
template <int U>
__global__ __launch_bounds__(256) void flat(Args p)
{
    const f4 *__restrict__ a = p.a[blockIdx.y];
    f4 *__restrict__ o = p.o[blockIdx.y];
    const uint32_t base = blockIdx.x * (U * 256u) + threadIdx.x;
    f4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = a[base + u * 256u];
#pragma unroll
    for (int u = 0; u < U; ++u) o[base + u * 256u] = work(x[u]);
}

// tile: block (bx, by) owns TWQ float4 columns x th rows; a thread owns column quad (tid % TWQ) and rows (tid / TWQ) + k * (256 / TWQ);
// RU rows in flight per trip, next trip requested before the current one is stored (as upsample_chain_tile does)
template <int RU, bool AHEAD>
__global__ __launch_bounds__(256) void tile(Args p, uint32_t twq, uint32_t th)
{
    const f4 *__restrict__ a = p.a[blockIdx.z];
    f4 *__restrict__ o = p.o[blockIdx.z];
    const uint32_t rgs = 256u / twq;
    const uint32_t cg = threadIdx.x % twq, rg = threadIdx.x / twq;
    const uint32_t col = blockIdx.x * twq + cg;
    const uint32_t y0 = blockIdx.y * th;
    f4 nxt[RU];
    if (AHEAD) {
#pragma unroll
        for (int u = 0; u < RU; ++u) nxt[u] = a[(size_t)(y0 + min(rg + u * rgs, th - 1)) * p.w4 + col];
    }
    for (uint32_t ty0 = rg; ty0 < th; ty0 += RU * rgs) {
        f4 x[RU];
        if (!AHEAD) {
#pragma unroll
            for (int u = 0; u < RU; ++u) nxt[u] = a[(size_t)(y0 + min(ty0 + u * rgs, th - 1)) * p.w4 + col];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) x[u] = nxt[u];
        if (AHEAD && ty0 + RU * rgs < th) {
#pragma unroll
            for (int u = 0; u < RU; ++u) nxt[u] = a[(size_t)(y0 + min(ty0 + RU * rgs + u * rgs, th - 1)) * p.w4 + col];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u)
            if (ty0 + u * rgs < th) o[(size_t)(y0 + ty0 + u * rgs) * p.w4 + col] = work(x[u]);
    }
}

// rows: block owns `th` whole rows; each trip the block covers RU * 256 consecutive float4 of a row (or of consecutive rows)
template <int RU>
__global__ __launch_bounds__(256) void rows(Args p, uint32_t th)
{
    const f4 *__restrict__ a = p.a[blockIdx.y];
    f4 *__restrict__ o = p.o[blockIdx.y];
    const size_t base = (size_t)blockIdx.x * th * p.w4;
    const uint32_t n = th * p.w4;
    for (uint32_t i = threadIdx.x; i < n; i += RU * 256u) {
        f4 x[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) x[u] = a[base + min(i + u * 256u, n - 1)];
#pragma unroll
        for (int u = 0; u < RU; ++u)
            if (i + u * 256u < n) o[base + i + u * 256u] = work(x[u]);
    }
}

// single trip: block owns 1024 columns x RU rows, every thread RU rows of one column quad, no loop
template <int RU>
__global__ __launch_bounds__(256) void tile1(Args p)
{
    const f4 *__restrict__ a = p.a[blockIdx.z];
    f4 *__restrict__ o = p.o[blockIdx.z];
    const uint32_t col = blockIdx.x * 256u + threadIdx.x;
    const uint32_t y0 = blockIdx.y * RU;
    f4 x[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) x[u] = a[(size_t)(y0 + u) * p.w4 + col];
#pragma unroll
    for (int u = 0; u < RU; ++u) o[(size_t)(y0 + u) * p.w4 + col] = work(x[u]);
}

// the same with a stand-in for the resampler's prologue between the loads and their use: a small dependent global read
// (the weights), a barrier, a second dependent read by 34 lanes with ~100 vector instructions and LDS writes (the vertical
// pass), a barrier, then three LDS reads per row feeding the result
template <int RU>
__global__ __launch_bounds__(256) void tile1_pre(Args p, const float *__restrict__ small)
{
    __shared__ float lds[RU * 140 + 64];
    const f4 *__restrict__ a = p.a[blockIdx.z];
    f4 *__restrict__ o = p.o[blockIdx.z];
    const uint32_t col = blockIdx.x * 256u + threadIdx.x;
    const uint32_t y0 = blockIdx.y * RU;
    f4 x[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) x[u] = a[(size_t)(y0 + u) * p.w4 + col];
    if (threadIdx.x < 64) lds[RU * 140 + threadIdx.x] = small[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 35) {
        const f4 *s4 = reinterpret_cast<const f4 *>(small) + 64 + (blockIdx.y & 63u) * 256u + blockIdx.x * 32u + threadIdx.x;
        f4 s0 = s4[0], s1 = s4[128], s2 = s4[256];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const float *w = lds + RU * 140 + 3 * u;
            f4 acc = { 0.f, 0.f, 0.f, 0.f };
            acc += s0 * w[0];
            acc += s1 * w[1];
            acc += s2 * w[2];
            *reinterpret_cast<f4 *>(lds + u * 140 + 4 * threadIdx.x) = acc;
        }
    }
    __syncthreads();
    const float *win = lds + (threadIdx.x >> 1);
    const float w0 = lds[RU * 140 + 24 + (threadIdx.x & 1u)], w1 = lds[RU * 140 + 26 + (threadIdx.x & 1u)], w2 = lds[RU * 140 + 28 + (threadIdx.x & 1u)];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const float *r = win + u * 140;
        const float t = (r[0] * w0 + r[1] * w1) + r[2] * w2;
        o[(size_t)(y0 + u) * p.w4 + col] = work(x[u]) + t;
    }
}

// tile1 with nontemporal loads and / or stores
template <int RU, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void tile1_nt(Args p)
{
    const f4 *__restrict__ a = p.a[blockIdx.z];
    f4 *__restrict__ o = p.o[blockIdx.z];
    const uint32_t col = blockIdx.x * 256u + threadIdx.x;
    const uint32_t y0 = blockIdx.y * RU;
    f4 x[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) x[u] = NTL ? __builtin_nontemporal_load(&a[(size_t)(y0 + u) * p.w4 + col]) : a[(size_t)(y0 + u) * p.w4 + col];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        if (NTS) __builtin_nontemporal_store(work(x[u]), &o[(size_t)(y0 + u) * p.w4 + col]);
        else o[(size_t)(y0 + u) * p.w4 + col] = work(x[u]);
    }
}

// 2-in-1-out (a single Mix node, config #1): one float4 per lane, flat
template <bool NTA, bool NTB, bool NTS>
__global__ __launch_bounds__(256) void add2(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const f4 x = NTA ? __builtin_nontemporal_load(&a[i]) : a[i];
    const f4 y = NTB ? __builtin_nontemporal_load(&b[i]) : b[i];
    if (NTS) __builtin_nontemporal_store(x + y, &o[i]);
    else o[i] = x + y;
}

int main()
{
    const uint32_t W = 4096, H = 4096, w4 = W / 4;
    const size_t bytes = (size_t)W * H * 4;
    Args p{};
    p.w4 = w4;
    p.h = H;
    for (int c = 0; c < 3; ++c) {
        CK(hipMalloc((void **)&p.a[c], bytes));
        CK(hipMalloc((void **)&p.o[c], bytes));
        CK(hipMemset((void *)p.a[c], 0x3c, bytes));
    }
    // something else to evict the Infinity Cache between runs is NOT used: the product runs back to back as well
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 20; ++i) launch();
        CK(hipDeviceSynchronize());
        std::vector<float> t;
        for (int i = 0; i < 60; ++i) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3f);
        }
        // and 100 back to back
        CK(hipEventRecord(e0));
        for (int i = 0; i < 100; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::sort(t.begin(), t.end());
        std::printf("%-44s single: min %.1f med %.1f us   back-to-back: %.1f us  = %.2f TB/s\n", name, t[0], t[t.size() / 2], ms * 10.f,
                    6.0 * bytes / (ms * 10e-6) / 1e12);
        CK(hipGetLastError());
    };
    const uint32_t n4 = w4 * H;
    float *small;
    CK(hipMalloc((void **)&small, 4 << 20));
    CK(hipMemset(small, 0, 4 << 20));
    {
        // config #1's traffic: 201 MB + 201 MB read, 201 MB written, as three 3-plane buffers; the result goes to one
        // buffer every launch ("same out") or to two in alternation ("2 outs": what a graph that still holds its previous
        // result does)
        const size_t big = 3 * bytes;
        f4 *A, *B, *O[2];
        CK(hipMalloc((void **)&A, big));
        CK(hipMalloc((void **)&B, big));
        CK(hipMalloc((void **)&O[0], big));
        CK(hipMalloc((void **)&O[1], big));
        CK(hipMemset(A, 0x3c, big));
        CK(hipMemset(B, 0x3c, big));
        const uint32_t nb = (uint32_t)(big / 16 / 256);
        int flip = 0;
#define ADD2(NAME, NA, NB, NS) \
        run(NAME " same out", [&] { add2<NA, NB, NS><<<nb, 256>>>(A, B, O[0]); }); \
        run(NAME " 2 outs", [&] { add2<NA, NB, NS><<<nb, 256>>>(A, B, O[flip++ & 1]); });
        ADD2("add2 plain           ", false, false, false)
        ADD2("add2 nt A,B          ", true, true, false)
        ADD2("add2 nt A,B,store    ", true, true, true)
        ADD2("add2 nt A,store      ", true, false, true)
        ADD2("add2 nt store        ", false, false, true)
        ADD2("add2 nt A            ", true, false, false)
    }
    {
        // warm: the same 201 MB of inputs every launch (what a re-evaluated graph does); cold: three input sets in rotation
        // (603 MB of inputs against the 256 MB Infinity Cache)
        Args q[3];
        for (int k = 0; k < 3; ++k) {
            q[k] = p;
            if (k)
                for (int c = 0; c < 3; ++c) {
                    CK(hipMalloc((void **)&q[k].a[c], bytes));
                    CK(hipMemset((void *)q[k].a[c], 0x3c, bytes));
                }
        }
        int rot = 0;
        run("tile1 1024 x 4            warm", [&] { tile1_nt<4, false, false><<<dim3(4, H / 4, 3), 256>>>(p); });
        run("tile1 1024 x 4 nt-store   warm", [&] { tile1_nt<4, false, true><<<dim3(4, H / 4, 3), 256>>>(p); });
        run("tile1 1024 x 4 nt-load    warm", [&] { tile1_nt<4, true, false><<<dim3(4, H / 4, 3), 256>>>(p); });
        run("tile1 1024 x 4 nt-both    warm", [&] { tile1_nt<4, true, true><<<dim3(4, H / 4, 3), 256>>>(p); });
        run("tile1 1024 x 4            cold", [&] { tile1_nt<4, false, false><<<dim3(4, H / 4, 3), 256>>>(q[rot++ % 3]); });
        run("tile1 1024 x 4 nt-store   cold", [&] { tile1_nt<4, false, true><<<dim3(4, H / 4, 3), 256>>>(q[rot++ % 3]); });
        run("tile1 1024 x 4 nt-load    cold", [&] { tile1_nt<4, true, false><<<dim3(4, H / 4, 3), 256>>>(q[rot++ % 3]); });
        run("tile1 1024 x 4 nt-both    cold", [&] { tile1_nt<4, true, true><<<dim3(4, H / 4, 3), 256>>>(q[rot++ % 3]); });
        run("tile1 1024 x 1 nt-store   cold", [&] { tile1_nt<1, false, true><<<dim3(4, H / 1, 3), 256>>>(q[rot++ % 3]); });
        run("tile1 1024 x 1 nt-both    cold", [&] { tile1_nt<1, true, true><<<dim3(4, H / 1, 3), 256>>>(q[rot++ % 3]); });
    }
    return 0;
    run("tile1 1024 x 1", [&] { tile1<1><<<dim3(4, H / 1, 3), 256>>>(p); });
    run("tile1 1024 x 2", [&] { tile1<2><<<dim3(4, H / 2, 3), 256>>>(p); });
    run("tile1 1024 x 4", [&] { tile1<4><<<dim3(4, H / 4, 3), 256>>>(p); });
    run("tile1 1024 x 8", [&] { tile1<8><<<dim3(4, H / 8, 3), 256>>>(p); });
    run("tile1 1024 x 16", [&] { tile1<16><<<dim3(4, H / 16, 3), 256>>>(p); });
    run("tile1_pre 1024 x 4", [&] { tile1_pre<4><<<dim3(4, H / 4, 3), 256>>>(p, small); });
    run("tile1_pre 1024 x 8", [&] { tile1_pre<8><<<dim3(4, H / 8, 3), 256>>>(p, small); });
    run("tile1_pre 1024 x 16", [&] { tile1_pre<16><<<dim3(4, H / 16, 3), 256>>>(p, small); });
    run("flat U=1", [&] { flat<1><<<dim3(n4 / 256, 3), 256>>>(p); });
    run("flat U=4", [&] { flat<4><<<dim3(n4 / 1024, 3), 256>>>(p); });
    for (uint32_t th : { 8u, 16u, 32u, 64u }) {
        char nm[96];
        std::snprintf(nm, sizeof nm, "tile 1024 x %u, 4 rows/trip, ahead", th);
        run(nm, [&] { tile<4, true><<<dim3(4, H / th, 3), 256>>>(p, 256, th); });
        std::snprintf(nm, sizeof nm, "tile 1024 x %u, 4 rows/trip", th);
        run(nm, [&] { tile<4, false><<<dim3(4, H / th, 3), 256>>>(p, 256, th); });
        std::snprintf(nm, sizeof nm, "tile 1024 x %u, 2 rows/trip, ahead", th);
        run(nm, [&] { tile<2, true><<<dim3(4, H / th, 3), 256>>>(p, 256, th); });
        std::snprintf(nm, sizeof nm, "tile 1024 x %u, 1 row/trip", th);
        run(nm, [&] { tile<1, false><<<dim3(4, H / th, 3), 256>>>(p, 256, th); });
    }
    for (uint32_t twq : { 128u, 64u }) {
        char nm[96];
        std::snprintf(nm, sizeof nm, "tile %u x 32, 4 rows/trip", twq * 4);
        run(nm, [&] { tile<4, false><<<dim3(w4 / twq, H / 32, 3), 256>>>(p, twq, 32); });
    }
    for (uint32_t th : { 1u, 2u, 4u, 8u, 16u }) {
        char nm[96];
        std::snprintf(nm, sizeof nm, "rows x %u, 1 KiB x 4 per trip", th);
        run(nm, [&] { rows<4><<<dim3(H / th, 3), 256>>>(p, th); });
        std::snprintf(nm, sizeof nm, "rows x %u, 1 KiB x 1 per trip", th);
        run(nm, [&] { rows<1><<<dim3(H / th, 3), 256>>>(p, th); });
    }
    return 0;
}
