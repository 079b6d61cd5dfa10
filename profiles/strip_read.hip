// How fast can column strips of a pitched plane be read?  Each wave reads a strip (64 lanes x VEC floats wide) down ROWS
// rows with PF row-loads in flight; strips tile a W x H f32 plane; successive wave-tiles overlap by OVL rows (the halo a
// resampling tile re-reads).  Prints GB/s of bytes requested.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 profiles/strip_read.hip -o /tmp/strip_read && /tmp/strip_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int VEC, int PF>
__global__ __launch_bounds__(256) void strip_kernel(const float *__restrict__ src, uint32_t pitch, uint32_t strips_x, uint32_t rows, uint32_t step,
                                                    uint32_t h, float *out)
{
    typedef float vt __attribute__((ext_vector_type(VEC)));
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t sx = wave % strips_x, sy = wave / strips_x;
    const uint32_t r0 = sy * step;
    if (r0 >= h) return;
    const uint32_t r1 = min(r0 + rows, h);
    const vt *col = reinterpret_cast<const vt *>(src + (size_t)sx * 64u * VEC) + lane;
    const uint32_t pv = pitch / VEC;
    vt acc = 0.0f;
    for (uint32_t r = r0; r < r1; r += PF) {
        vt p[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) p[u] = col[(size_t)min(r + u, r1 - 1u) * pv];
#pragma unroll
        for (int u = 0; u < PF; ++u) acc += p[u];
    }
    float s = 0.0f;
    for (int i = 0; i < VEC; ++i) s += acc[i];
    if (s == 123.456f) out[0] = s;
}

template <int VEC, int PF>
static void run(const float *src, uint32_t w, uint32_t h, uint32_t rows, uint32_t ovl, float *out)
{
    const uint32_t strips_x = w / (64u * VEC), step = rows - ovl, strips_y = (h + step - 1) / step;
    const uint32_t waves = strips_x * strips_y, blocks = (waves + 3) / 4;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) strip_kernel<VEC, PF><<<blocks, 256>>>(src, w, strips_x, rows, step, h, out);
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) strip_kernel<VEC, PF><<<blocks, 256>>>(src, w, strips_x, rows, step, h, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    const double req = (double)strips_x * 64 * VEC * 4 * (double)strips_y * rows;
    std::printf("VEC=%d PF=%2d rows=%4u ovl=%3u waves=%6u : %7.1f us, %6.0f GB/s requested, %6.0f GB/s of the plane\n", VEC, PF, rows, ovl, waves, us,
                req / us * 1e-3, (double)w * h * 4 / us * 1e-3);
}

int main()
{
    const uint32_t w = 4096, h = 4096;
    float *src, *out;
    CK(hipMalloc((void **)&src, (size_t)w * h * 4));
    CK(hipMalloc((void **)&out, 64));
    CK(hipMemset(src, 0, (size_t)w * h * 4));
    for (uint32_t rows : { 16u, 37u, 85u, 277u, 4096u }) {
        run<4, 4>(src, w, h, rows, 0, out);
        run<4, 8>(src, w, h, rows, 0, out);
        run<2, 8>(src, w, h, rows, 0, out);
        run<1, 8>(src, w, h, rows, 0, out);
        run<1, 16>(src, w, h, rows, 0, out);
    }
    std::printf("-- with the halo a 16-row Lanczos3 4:1 tile re-reads (85 rows, 21 shared with the next tile)\n");
    run<4, 4>(src, w, h, 85, 21, out);
    run<4, 8>(src, w, h, 85, 21, out);
    run<2, 8>(src, w, h, 85, 21, out);
    run<1, 8>(src, w, h, 85, 21, out);
    std::printf("-- 4-row groups (37 rows, 21 shared)\n");
    run<4, 4>(src, w, h, 37, 21, out);
    run<4, 8>(src, w, h, 37, 21, out);
    return 0;
}
