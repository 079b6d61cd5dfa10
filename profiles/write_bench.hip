// What does the memory system take as WRITES?  A standalone HIP program (not part of the product): a 64 MiB plane filled with a
// constant by kernels that differ in the store (plain / nontemporal), the bytes a thread stores in a row, the number of workgroups
// and which part of the plane an XCD writes.  The up-sampling kernels and fill_kernel are write-bound; this is their ceiling.
//   hipcc --offload-arch=gfx950 -O3 profiles/write_bench.hip -o gpurun_out/write_bench && gpurun_out/write_bench
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

typedef float f4 __attribute__((ext_vector_type(4)));

// grid-stride: a workgroup's 256 threads store 4 KiB contiguous, then the workgroup moves on by gridDim * 4 KiB
template <bool NT>
__global__ __launch_bounds__(256) void fill_stride(f4 *dst, uint32_t n4, float v)
{
    const f4 val = { v, v, v, v };
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        if (NT) __builtin_nontemporal_store(val, dst + i);
        else dst[i] = val;
    }
}

// block-contiguous: workgroup b stores U * 4 KiB contiguous; XCD != 0: workgroup id % 8 = XCD k fills the k-th eighth of the plane
template <int U, bool NT, bool XCD>
__global__ __launch_bounds__(256) void fill_block(f4 *dst, uint32_t n4, float v)
{
    const f4 val = { v, v, v, v };
    uint32_t b = blockIdx.x;
    if (XCD) b = (b & 7u) * (gridDim.x / 8u) + (b >> 3);
    const uint32_t base = b * (U * 256u) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t i = base + u * 256u;
        if (i < n4) {
            if (NT) __builtin_nontemporal_store(val, dst + i);
            else dst[i] = val;
        }
    }
}

template <typename F>
static void timeit(const char *what, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 30;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    const double us = ms * 1000.0 / reps;
    std::printf("%-64s %6.1f us  %5.2f TB/s\n", what, us, 67108864.0 / us / 1e6);
}

int main()
{
    const uint32_t n4 = 4096u * 4096u / 4u;
    f4 *dst;
    CK(hipMalloc(&dst, (size_t)n4 * 16));
    for (unsigned blocks : { 1024u, 2048u, 4096u, 8192u, 16384u }) {
        char name[96];
        std::snprintf(name, sizeof name, "grid-stride, plain stores, %u workgroups", blocks);
        timeit(name, [&] { fill_stride<false><<<blocks, 256>>>(dst, n4, 0.5f); });
        std::snprintf(name, sizeof name, "grid-stride, nontemporal stores, %u workgroups", blocks);
        timeit(name, [&] { fill_stride<true><<<blocks, 256>>>(dst, n4, 0.5f); });
    }
#define BLK(U, NT, XCD)                                                                                                        \
    {                                                                                                                          \
        const unsigned blocks = (n4 + U * 256u - 1) / (U * 256u);                                                              \
        char name[96];                                                                                                         \
        std::snprintf(name, sizeof name, "block-contiguous %d x 4 KiB, %s stores%s, %u workgroups", U, NT ? "nontemporal" : "plain", \
                      XCD ? ", an eighth of the plane per XCD" : "", blocks);                                                   \
        timeit(name, [&] { fill_block<U, NT, XCD><<<blocks, 256>>>(dst, n4, 0.5f); });                                         \
    }
    BLK(1, false, false) BLK(1, true, false) BLK(2, false, false) BLK(2, true, false) BLK(4, false, false) BLK(4, true, false)
    BLK(8, false, false) BLK(8, true, false) BLK(4, false, true) BLK(4, true, true) BLK(8, false, true) BLK(8, true, true)
    BLK(16, false, false) BLK(16, true, true)
    return 0;
}
