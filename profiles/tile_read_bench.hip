// What does the memory system give the band waves' ACCESS PATTERN, with the arithmetic taken away?
// Not part of the product: a standalone HIP program.  Every wave is a (strip, band) job as in resize_poly_kernel: it walks down
// `trips` trips of RT source rows, a lane reading LW floats of each row, NB trips in flight, `alu` packed multiply + add pairs per row
// per lane in between (0: a plain sum).  Prints the launch time and the fetched bytes per second for a list of geometries.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/tile_read_bench.hip -o gpurun_out/tile_read_bench && gpurun_out/tile_read_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
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
    uint32_t alu;         // packed multiply + add pairs per row per lane (a template parameter; here for the printout)
    uint32_t mis;         // floats added to every window's first column (a multiple of LW)
    uint32_t wps;         // waves side by side on one strip window (n_strips counts the waves' part-strips)
    uint32_t waves_per_wg;
    uint32_t order;       // 0: job = band * n_strips + strip (neighbouring strips in a workgroup); 1: job = strip * n_bands + band
                          // (neighbouring bands in a workgroup); 2 / 3: the same with the workgroups dealt to the XCDs in eighths
};

template <int LW, int RT, int NB, int ALU>
__global__ __launch_bounds__(256) void tile_read(const float *__restrict__ src, float *__restrict__ out, Geo g)
{
    typedef float fv __attribute__((ext_vector_type(LW)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t wg = blockIdx.x;
    if (g.order >= 2u) {  // workgroup id % 8 is the XCD: XCD k takes the k-th eighth of the workgroups in the order below
        const uint32_t per = (gridDim.x + 7u) / 8u;
        wg = (wg & 7u) * per + (wg >> 3);
        if (wg >= gridDim.x) return;
    }
    const uint32_t job = wg * g.waves_per_wg + wave;
    if (job >= g.n_strips * g.n_bands) return;
    const bool vertical = g.order == 1u || g.order == 3u;  // a workgroup's waves: bands of one strip (else: strips of one band)
    const uint32_t band = vertical ? job % g.n_bands : job / g.n_strips;
    const uint32_t strip = vertical ? job / g.n_bands : job % g.n_strips;
    // wps waves share a strip's window side by side (wps * 64 * LW columns for strip_new new ones)
    const uint32_t wps = g.wps ? g.wps : 1u;
    const uint32_t c0 = (std::min((strip / wps) * g.strip_new, g.size - 64u * LW * wps) & ~3u) + (strip % wps) * 64u * LW;
    const uint32_t r0 = band * g.band_new;
    const uint32_t c0m = std::min(c0 + g.mis, g.size - 64u * LW);  // (mis: the window does not start on a cache line)
    const fv *col = reinterpret_cast<const fv *>(src + c0m) + lane;
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
                    if constexpr (ALU == 0) acc[u % 6] += v;
#pragma unroll
                    for (int k = 0; k < ALU; ++k) acc[(u + k) % 6] += v * w;  // (-ffp-contract=off: a multiply and an add)
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

template <int LW, int RT, int NB, int ALU>
static float run(const float *src, float *out, const Geo &g, int reps)
{
    const uint32_t jobs = g.n_strips * g.n_bands;
    dim3 grid((jobs + g.waves_per_wg - 1) / g.waves_per_wg);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) tile_read<LW, RT, NB, ALU><<<grid, 64 * g.waves_per_wg>>>(src, out, g);
    CK(hipDeviceSynchronize());
    // TILE_READ_GAP_US=n: the host waits for every launch and then n microseconds more before the next one (how a caller that
    // does other work between resizes sees the kernel; the times printed then include the gaps -- read them from a kernel trace)
    static const int gap_us = std::getenv("TILE_READ_GAP_US") ? std::atoi(std::getenv("TILE_READ_GAP_US")) : -1;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) {
        tile_read<LW, RT, NB, ALU><<<grid, 64 * g.waves_per_wg>>>(src, out, g);
        if (gap_us >= 0) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < gap_us) {}
        }
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1000.0f / reps;
}

template <int LW, int RT, int ALU>
static float run_nb(int nb, const float *src, float *out, const Geo &g, int reps)
{
    switch (nb) {
    case 1: return run<LW, RT, 1, ALU>(src, out, g, reps);
    case 2: return run<LW, RT, 2, ALU>(src, out, g, reps);
    default: return run<LW, RT, 3, ALU>(src, out, g, reps);
    }
}

template <int LW, int RT>
static float run_alu(int alu, int nb, const float *src, float *out, const Geo &g, int reps)
{
    switch (alu) {
    case 0: return run_nb<LW, RT, 0>(nb, src, out, g, reps);
    case 3: return run_nb<LW, RT, 3>(nb, src, out, g, reps);
    case 6: return run_nb<LW, RT, 6>(nb, src, out, g, reps);
    default: return run_nb<LW, RT, 12>(nb, src, out, g, reps);
    }
}

static float run_any(int lw, int rt, int alu, int nb, const float *src, float *out, const Geo &g, int reps)
{
    if (lw == 4) return rt == 8 ? run_alu<4, 8>(alu, nb, src, out, g, reps) : rt == 4 ? run_alu<4, 4>(alu, nb, src, out, g, reps) : run_alu<4, 2>(alu, nb, src, out, g, reps);
    if (lw == 2) return rt == 8 ? run_alu<2, 8>(alu, nb, src, out, g, reps) : rt == 4 ? run_alu<2, 4>(alu, nb, src, out, g, reps) : run_alu<2, 2>(alu, nb, src, out, g, reps);
    return rt == 8 ? run_alu<1, 8>(alu, nb, src, out, g, reps) : rt == 4 ? run_alu<1, 4>(alu, nb, src, out, g, reps) : run_alu<1, 2>(alu, nb, src, out, g, reps);
}

int main(int argc, char **argv)
{
    const uint32_t size = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 4096u;
    const int reps = 30;
    float *src, *out;
    CK(hipMalloc(&src, (size_t)size * size * 4));
    CK(hipMalloc(&out, (size_t)64 << 20));
    if (argc > 2 && std::strcmp(argv[2], "zeros") == 0) {
        CK(hipMemset(src, 0, (size_t)size * size * 4));  // (all-zero planes read measurably faster than real data: see profiles/r04_poly_weights.md)
    } else {
        std::vector<float> h((size_t)size * size);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (float &v : h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            v = (float)(x >> 40) * (1.0f / 16777216.0f);
        }
        CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    std::printf("# source %u^2 f32, %s (%.1f MB algorithmic); columns: lane floats, rows per trip, trips in flight, new cols per strip, rows per band (output rows at this ratio),\n"
                "# ages, alu per row, waves per workgroup, order -> waves, workgroups, fetched MB, us, fetched TB/s, algorithmic TB/s\n", size, argc > 2 ? argv[2] : "random values", size * (double)size * 4 / 1e6);
    struct Case { int lw, rt, nb; uint32_t strip_new, rows, ages, alu, wpw, order, wps, mis; };
    std::vector<Case> cases;
    // the shipped geometry at ratio 8 (Gaussian: 6 ages; windows of 256 columns for 192 new ones, bands of 12 rows), loads only, by trips in flight
    for (int nb : { 1, 2, 3 }) cases.push_back({ 4, 8, nb, 192, 12, 6, 0, 4, 0, 1 });
    // band height
    for (uint32_t rows : { 8u, 16u, 24u, 32u, 64u }) cases.push_back({ 4, 8, 2, 192, rows, 6, 0, 4, 0, 1 });
    // no overlap at all (what the pattern itself costs)
    for (uint32_t rows : { 8u, 12u, 16u, 32u }) cases.push_back({ 4, 8, 2, 256, rows, 1, 0, 4, 0, 1 });
    // narrower lanes with windows of their own (the first resize_poly2_kernel: 128-column windows for 80 new ones)
    for (int nb : { 1, 3 })
        for (uint32_t rows : { 12u, 24u }) cases.push_back({ 2, 8, nb, 80, rows, 6, 0, 4, 0, 1 });
    // ratio 4 and 2 (Lanczos3: 6 ages): 232 new columns of 256, 244
    for (uint32_t rows : { 12u, 24u, 48u }) cases.push_back({ 4, 4, 2, 232, rows, 6, 0, 4, 0, 1 });
    for (uint32_t rows : { 12u, 24u, 48u, 96u }) cases.push_back({ 4, 2, 2, 244, rows, 6, 0, 4, 0, 1 });
    // which jobs share a workgroup and an XCD (with the arithmetic of resize_poly_kernel)
    for (uint32_t ord : { 0u, 1u, 2u, 3u }) cases.push_back({ 4, 8, 1, 192, 12, 6, 12, 4, ord, 1 });
    for (uint32_t ord : { 0u, 1u, 2u, 3u }) cases.push_back({ 4, 4, 1, 232, 12, 6, 12, 4, ord, 1 });
    for (uint32_t ord : { 0u, 1u, 2u, 3u }) cases.push_back({ 4, 2, 1, 244, 12, 6, 12, 4, ord, 1 });
    for (uint32_t ord : { 0u, 2u }) cases.push_back({ 2, 8, 1, 192, 12, 6, 6, 4, ord, 2 });
    // windows that do not start on a 128-byte line (the kernels' windows start at a strip's first tap rounded down to 16 bytes)
    for (uint32_t mis : { 4u, 8u, 16u, 28u }) cases.push_back({ 4, 8, 1, 192, 12, 6, 12, 4, 2, 1, mis });
    for (uint32_t mis : { 4u, 16u }) cases.push_back({ 2, 8, 1, 192, 12, 6, 6, 4, 2, 2, mis });
    for (uint32_t mis : { 4u, 16u }) cases.push_back({ 4, 4, 1, 232, 12, 6, 12, 4, 2, 1, mis });
    for (uint32_t mis : { 4u, 16u }) cases.push_back({ 4, 2, 1, 244, 12, 6, 12, 4, 2, 1, mis });
    // strips whose width is not a multiple of 32 columns: every second window starts off a line
    for (uint32_t sn : { 216u, 208u, 200u, 224u }) cases.push_back({ 4, 8, 1, sn, 12, 6, 12, 4, 2, 1, 0 });
    // ---- with the vertical pass's arithmetic (packed multiply + add pairs per row and lane: 6 ages x 2 halves = 12 for 16-byte
    // lanes, 6 for 8-byte lanes, 3 + for 4-byte lanes where half of every packed operation is idle) ----
    for (int nb : { 1, 2, 3 }) cases.push_back({ 4, 8, nb, 192, 12, 6, 12, 4, 0, 1 });   // resize_poly_kernel
    for (int nb : { 1, 2, 3 }) cases.push_back({ 2, 8, nb, 192, 12, 6, 6, 4, 0, 2 });    // resize_poly2_kernel: two waves per window
    for (int nb : { 1, 2, 3 }) cases.push_back({ 1, 8, nb, 192, 12, 6, 6, 4, 0, 4 });    // four waves per window, plain operations
    for (int nb : { 1, 2, 3 }) cases.push_back({ 1, 8, nb, 192, 12, 6, 3, 4, 0, 4 });    // ... if they could be packed
    for (uint32_t rows : { 8u, 16u, 24u }) cases.push_back({ 2, 8, 2, 192, rows, 6, 6, 4, 0, 2 });
    for (uint32_t rows : { 16u, 24u, 32u }) cases.push_back({ 1, 8, 2, 192, rows, 6, 6, 4, 0, 4 });
    // arithmetic alone (no overlap, so few bytes): what the instruction stream costs these wave counts
    cases.push_back({ 4, 8, 1, 256, 12, 1, 12, 4, 0, 1 });
    cases.push_back({ 2, 8, 1, 256, 12, 1, 6, 4, 0, 2 });
    // ratio 4: poly, pairs
    for (int nb : { 1, 2 }) cases.push_back({ 4, 4, nb, 232, 12, 6, 12, 4, 0, 1 });
    for (int nb : { 1, 2 }) cases.push_back({ 2, 4, nb, 232, 12, 6, 6, 4, 0, 2 });
    for (uint32_t rows : { 24u }) cases.push_back({ 4, 4, 2, 232, rows, 6, 12, 4, 0, 1 });
    for (const Case &c : cases) {
        Geo g{};
        g.size = size;
        g.strip_new = c.strip_new;
        g.wps = c.wps;
        g.mis = c.mis;
        g.n_strips = (size + c.strip_new - 1) / c.strip_new * c.wps;
        const uint32_t out_rows = size / c.rt;
        g.n_bands = (out_rows + c.rows - 1) / c.rows;
        g.band_new = c.rows * c.rt;
        g.trips = c.rows + c.ages - 1;
        g.alu = c.alu;
        g.waves_per_wg = c.wpw;
        g.order = c.order;
        const uint32_t jobs = g.n_strips * g.n_bands;
        if ((size_t)jobs * 64 * 4 > ((size_t)64 << 20)) continue;
        const float us = run_any(c.lw, c.rt, (int)c.alu, c.nb, src, out, g, reps);
        const double fetched = (double)jobs * g.trips * c.rt * 64.0 * c.lw * 4.0;
        std::printf("lw=%d rt=%d nb=%d new=%3u rows=%2u ages=%u alu=%2u wpw=%u wps=%u ord=%u mis=%2u -> %5u waves %5u wgs  %6.1f MB  %6.1f us  %5.2f TB/s fetched  %5.2f TB/s algorithmic\n", c.lw,
                    c.rt, c.nb, c.strip_new, c.rows, c.ages, c.alu, c.wpw, c.wps, c.order, c.mis, jobs, (jobs + c.wpw - 1) / c.wpw, fetched / 1e6, us, fetched / us / 1e6,
                    size * (double)size * 4 / us / 1e6);
        std::fflush(stdout);
    }
    return 0;
}
