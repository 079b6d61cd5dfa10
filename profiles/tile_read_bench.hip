// What does the memory system give the band waves' ACCESS PATTERN, with the arithmetic taken away?
// Not part of the product: a standalone HIP program.  Every wave is a (strip, band) job as in resize_poly_kernel: it walks down
// `trips` trips of RT source rows, a lane reading LW floats of each row, NB trips in flight, `alu` packed multiply-adds per row
// per lane in between (0: a plain sum).  Prints the launch time and the fetched bytes per second for a list of geometries.
//   hipcc --offload-arch=gfx950 -O3 profiles/tile_read_bench.hip -o gpurun_out/tile_read_bench && gpurun_out/tile_read_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

struct Geo {
    uint32_t size;        // source is size x size floats
    uint32_t n_strips;    // strips per row
    uint32_t strip_new;   // new columns per strip (its window is 64 * LW columns)
    uint32_t n_bands;
    uint32_t band_new;    // new source rows per band
    uint32_t trips;       // trips of RT rows per band (band_new / RT + overlap)
    uint32_t alu;         // packed multiply-adds per row per lane
    uint32_t waves_per_wg;
    uint32_t order;       // 0: job = band * n_strips + strip (neighbouring strips in a workgroup); 1: job = strip * n_bands + band
};

template <int LW, int RT, int NB>
__global__ __launch_bounds__(256) void tile_read(const float *__restrict__ src, float *__restrict__ out, Geo g)
{
    typedef float fv __attribute__((ext_vector_type(LW)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t job = blockIdx.x * g.waves_per_wg + wave;
    if (job >= g.n_strips * g.n_bands) return;
    const uint32_t band = g.order ? job % g.n_bands : job / g.n_strips;
    const uint32_t strip = g.order ? job / g.n_bands : job % g.n_strips;
    const uint32_t c0 = std::min(strip * g.strip_new, g.size - 64u * LW) & ~3u;
    const uint32_t r0 = band * g.band_new;
    const fv *col = reinterpret_cast<const fv *>(src + c0) + lane;
    const uint32_t pitch = g.size / LW;
    auto row = [&](uint32_t t, int u) { return (size_t)std::min(r0 + t * RT + u, g.size - 1u) * pitch; };
    fv pb[NB][RT];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int u = 0; u < RT; ++u) pb[b][u] = col[row(std::min((uint32_t)b, g.trips - 1u), u)];
    f2 acc[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[a] = f2{ 0.0f, 0.0f };
    const f2 w = f2{ 1.0f + 1e-7f * lane, 1.0f - 1e-7f * lane };
    for (uint32_t cb = 0; cb < g.trips; cb += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const uint32_t c = cb + b;
            fv(&p)[RT] = pb[b];
            if (c < g.trips) {
#pragma unroll
                for (int u = 0; u < RT; ++u) {
                    f2 v;
                    if constexpr (LW == 4) v = f2{ p[u].x + p[u].z, p[u].y + p[u].w };
                    else if constexpr (LW == 2) v = f2{ p[u].x, p[u].y };
                    else v = f2{ p[u][0], p[u][0] };
                    acc[u % 6] += v;
                    for (uint32_t k = 0; k < g.alu; ++k) acc[(u + k) % 6] = acc[(u + k) % 6] * w + v;
                }
            }
            const uint32_t cn = std::min(c + (uint32_t)NB, g.trips - 1u);
#pragma unroll
            for (int u = 0; u < RT; ++u) {
                if constexpr (LW > 1) asm volatile("" : "+v"(p[u]));
                p[u] = col[row(cn, u)];
            }
        }
    }
    f2 s = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5];
    out[(size_t)job * 64u + lane] = s.x + s.y;
}

template <int LW, int RT, int NB>
static float run(const float *src, float *out, const Geo &g, int reps)
{
    const uint32_t jobs = g.n_strips * g.n_bands;
    dim3 grid((jobs + g.waves_per_wg - 1) / g.waves_per_wg);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) tile_read<LW, RT, NB><<<grid, 64 * g.waves_per_wg>>>(src, out, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) tile_read<LW, RT, NB><<<grid, 64 * g.waves_per_wg>>>(src, out, g);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1000.0f / reps;
}

template <int LW, int RT>
static float run_nb(int nb, const float *src, float *out, const Geo &g, int reps)
{
    switch (nb) {
    case 1: return run<LW, RT, 1>(src, out, g, reps);
    case 2: return run<LW, RT, 2>(src, out, g, reps);
    case 3: return run<LW, RT, 3>(src, out, g, reps);
    default: return run<LW, RT, 4>(src, out, g, reps);
    }
}

static float run_any(int lw, int rt, int nb, const float *src, float *out, const Geo &g, int reps)
{
    if (lw == 4) return rt == 8 ? run_nb<4, 8>(nb, src, out, g, reps) : rt == 4 ? run_nb<4, 4>(nb, src, out, g, reps) : run_nb<4, 2>(nb, src, out, g, reps);
    if (lw == 2) return rt == 8 ? run_nb<2, 8>(nb, src, out, g, reps) : rt == 4 ? run_nb<2, 4>(nb, src, out, g, reps) : run_nb<2, 2>(nb, src, out, g, reps);
    return rt == 8 ? run_nb<1, 8>(nb, src, out, g, reps) : rt == 4 ? run_nb<1, 4>(nb, src, out, g, reps) : run_nb<1, 2>(nb, src, out, g, reps);
}

int main(int argc, char **argv)
{
    const uint32_t size = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 4096u;
    const int reps = 30;
    float *src, *out;
    CK(hipMalloc(&src, (size_t)size * size * 4));
    CK(hipMalloc(&out, (size_t)64 << 20));
    CK(hipMemset(src, 0, (size_t)size * size * 4));
    std::printf("# source %u^2 f32 (%.1f MB algorithmic); columns: lane floats, rows per trip, trips in flight, new cols per strip, rows per band (output rows at this ratio),\n"
                "# ages, alu per row, waves per workgroup, order -> waves, workgroups, fetched MB, us, fetched TB/s, algorithmic TB/s\n", size, size * (double)size * 4 / 1e6);
    struct Case { int lw, rt, nb; uint32_t strip_new, rows, ages, alu, wpw, order; };
    std::vector<Case> cases;
    // the shipped geometry at ratio 8 (Gaussian: 6 ages; windows of 256 columns for 192 new ones, bands of 12 rows), and what prefetch depth does to it
    for (int nb : { 1, 2, 3 })
        for (uint32_t alu : { 0u, 6u }) cases.push_back({ 4, 8, nb, 192, 12, 6, alu, 4, 0 });
    // band height
    for (uint32_t rows : { 8u, 16u, 24u, 32u, 64u }) cases.push_back({ 4, 8, 2, 192, rows, 6, 0, 4, 0 });
    // no overlap at all (what the pattern itself costs)
    for (uint32_t rows : { 8u, 12u, 16u, 32u }) cases.push_back({ 4, 8, 2, 256, rows, 1, 0, 4, 0 });
    // narrower lanes (resize_poly2_kernel's geometry: 128-column windows for 80 new ones)
    for (int nb : { 1, 3 })
        for (uint32_t rows : { 12u, 24u }) cases.push_back({ 2, 8, nb, 80, rows, 6, 0, 4, 0 });
    cases.push_back({ 2, 8, 3, 128, 12, 1, 0, 4, 0 });
    cases.push_back({ 1, 8, 3, 64, 12, 1, 0, 4, 0 });
    // one wave per workgroup / two (placement), and strips of one band apart
    for (uint32_t wpw : { 1u, 2u }) cases.push_back({ 4, 8, 2, 192, 12, 6, 0, wpw, 0 });
    cases.push_back({ 4, 8, 2, 192, 12, 6, 0, 4, 1 });
    // ratio 4 and 2 (Lanczos3: 6 ages): 232 new columns of 256, 244
    for (int nb : { 1, 2 }) cases.push_back({ 4, 4, nb, 232, 12, 6, 0, 4, 0 });
    for (uint32_t rows : { 24u, 48u }) cases.push_back({ 4, 4, 2, 232, rows, 6, 0, 4, 0 });
    for (int nb : { 1, 2 }) cases.push_back({ 4, 2, nb, 244, 12, 6, 0, 4, 0 });
    for (uint32_t rows : { 24u, 48u, 96u }) cases.push_back({ 4, 2, 2, 244, rows, 6, 0, 4, 0 });
    for (const Case &c : cases) {
        Geo g{};
        g.size = size;
        g.strip_new = c.strip_new;
        g.n_strips = (size + c.strip_new - 1) / c.strip_new;
        const uint32_t out_rows = size / c.rt;
        g.n_bands = (out_rows + c.rows - 1) / c.rows;
        g.band_new = c.rows * c.rt;
        g.trips = c.rows + c.ages - 1;
        g.alu = c.alu;
        g.waves_per_wg = c.wpw;
        g.order = c.order;
        const uint32_t jobs = g.n_strips * g.n_bands;
        if ((size_t)jobs * 64 * 4 > ((size_t)64 << 20)) continue;
        const float us = run_any(c.lw, c.rt, c.nb, src, out, g, reps);
        const double fetched = (double)jobs * g.trips * c.rt * 64.0 * c.lw * 4.0;
        std::printf("lw=%d rt=%d nb=%d new=%3u rows=%2u ages=%u alu=%u wpw=%u ord=%u -> %5u waves %5u wgs  %6.1f MB  %6.1f us  %5.2f TB/s fetched  %5.2f TB/s algorithmic\n", c.lw,
                    c.rt, c.nb, c.strip_new, c.rows, c.ages, c.alu, c.wpw, c.order, jobs, (jobs + c.wpw - 1) / c.wpw, fetched / 1e6, us, fetched / us / 1e6,
                    size * (double)size * 4 / us / 1e6);
        std::fflush(stdout);
    }
    return 0;
}
