// Down-sampling on both axes (windows of 4 taps or more each way; image::imageops::resize of crate image 0.24.0 as called from
// src/shared.rs:159-199), third form: every WAVE is a job of its own -- no workgroup barrier, no staging shared between waves.
//
// What bounded resize_down_kernel (profiles/r03_down_lds_experiment.txt) was not traffic but a wave's chain of dependent
// waits: four trips of loads, and inside every trip scalar weight loads whose addresses depend on window tests (up to 48
// load -> wait -> use round trips per wave), on 25 of 64 lanes, at 3-4 workgroups per CU.  Here:
//   vertical pass    a wave owns 4 adjacent output rows and 64 source column quads.  The host lays the rows' weights out
//                    DENSELY per source row (resize.cpp, down2_build): record (group, chunk) = first source row, a 64-bit
//                    presence mask and w[16 source rows][4 output rows].  The wave fetches the record with scalar loads at
//                    addresses that depend on nothing but its row group, issues its 16 row loads unconditionally (clamped
//                    to the group's last window row) and then runs straight through: tap (u, k) is one scalar bit test
//                    around two packed multiplies and two packed adds.  Absent taps are skipped, not multiplied by zero:
//                    the sums receive exactly the reference's terms in the reference's order.
//   transposition    the four sums of a column go to the wave's own 4 KB of LDS as ONE 16-byte entry (rows 0-3 of that
//                    column), so that
//   horizontal pass  a lane owns an output column (HC of them, 64 apart) for all four rows: one ds_read_b128 per tap feeds
//                    four sums; its weights sit in registers, loaded as 16-byte quads from a copy of the horizontal table
//                    whose rows are padded to a multiple of four.  Taps past a column's own count are replaced by -0.0,
//                    which leaves every sum unchanged (as in resize_down_hrows).
// A strip is as wide as 64 source quads allow (178 columns for Lanczos3 at 4096 -> 3000): every lane of the vertical pass
// has a quad, and 16.5 KB of LDS per workgroup carry 2 848 results instead of 1 024.
#include "kc_internal.hpp"

namespace kc {
namespace {
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// image::math::utils::clamp(t, 0, 1) for a t that is never -0.0 (a sum that started at +0.0): the IEEE-754-2019 maximum / minimum
// (v_maximum3_f32 / v_minimum3_f32) pass a NaN through, which is what the reference's two comparisons do (as up_clamp01).
__device__ __forceinline__ f4 d2_clamp(f4 t)
{
    const f4 zero = { 0.0f, 0.0f, 0.0f, 0.0f }, one = { 1.0f, 1.0f, 1.0f, 1.0f };
    return __builtin_elementwise_minimum(__builtin_elementwise_maximum(t, zero), one);
}

typedef uint32_t u16v __attribute__((ext_vector_type(16)));
typedef const u16v __attribute__((address_space(4))) *d2_const_u16v;
typedef const uint32_t __attribute__((address_space(4))) *d2_const_u32;
__device__ __forceinline__ d2_const_u32 d2_const(const uint32_t *p)
{
    return (d2_const_u32)(uintptr_t)p;  // kernel-lifetime constants: scalar loads
}
}  // namespace

// The columns' weights are loaded AFTER the vertical pass (4-5 waves per SIMD).  Loading them behind the row loads, in flight
// during the arithmetic, costs a wave per SIMD and measured the same or slower (28.5 against 27.7 us on Lanczos3 4096^2 ->
// 3000^2), and so did loading them half way through the arithmetic (profiles/r03_down2_variants.txt).
// MODE 0: every row group has a single chunk (ratios below about 1.6); 1: several, eight rows at a time, pipelined
template <int HC, int NW4, int MODE>
__global__ __launch_bounds__(256) void resize_down2_kernel(const ResizePlanes P, const Down2Args A)
{
    __shared__ f4 T[4][KC_DOWN2_SLOTS];  // wave-private: T[wave][source column of the strip] = (row 0, 1, 2, 3)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    // Which tile: workgroups go to the 8 XCDs in turn (id % 8), each XCD with an L2 of its own, and vertically adjacent tiles
    // share 9 of their ~30 source rows.  With xcd_per set the grid is one-dimensional and XCD k works through the k-th
    // eighth of the tiles in strip-major order, so that the rows two neighbours share are fetched into one L2, once.
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    uint32_t g;  // row group: output rows 4 g .. 4 g + 3
    if (A.by_rows) {
        const uint32_t tile = (blockIdx.x & 7u) * A.xcd_per + (blockIdx.x >> 3);
        const uint32_t job = tile * 4u + wave;
        if (job >= A.n_tiles) return;  // (n_tiles: jobs here; no barrier anywhere in this kernel)
        g = __umulhi(job, A.gy_magic);  // job / strips
        bx = job - g * A.gy;
    } else
    if (A.xcd_per) {
        const uint32_t tile = (blockIdx.x & 7u) * A.xcd_per + (blockIdx.x >> 3);
        if (tile >= A.n_tiles) return;
        bx = __umulhi(tile, A.gy_magic);  // tile / gy (host-checked to be exact for every tile)
        by = tile - bx * A.gy;
    }
    if (!A.by_rows) g = by * 4u + wave;
    if (4u * g >= A.dh) return;         // (no barrier anywhere in this kernel)
    const uint32_t x0 = bx * A.tile_w, x1 = min(x0 + A.tile_w, A.dw);
    // the strip's first source column (a multiple of 4) and its width in quads (<= 64), and the group's first record: two
    // scalar loads whose addresses depend on the block and wave index only, in flight together
    const d2_const_u32 strip = d2_const(A.strips) + 2u * bx;
    const d2_const_u32 rec0 = d2_const(A.vrec) + (size_t)g * A.nc * KC_DOWN2_REC;
    const uint32_t c0 = strip[0], nq = strip[1];
    const uint32_t z = blockIdx.z;
    // rows are addressed as (uniform row base) + (this lane's quad): the row bases stay in scalar registers
    const float *src0 = P.src[z] + c0;
    const uint32_t spitch = P.spitch[z];
    const uint32_t qi = min(lane, nq - 1u);

    // ---- this lane's output columns: window start, tap count, weights ----
    uint32_t h0[HC], n[HC];
    f4 w[HC][NW4];
    auto columns = [&]() {
#pragma unroll
        for (int c = 0; c < HC; ++c) {
            const uint32_t x = min(x0 + lane + 64u * c, x1 - 1u);
            h0[c] = A.hleft[x] - c0;
            n[c] = A.hcount[x];
            const f4 *wrow = reinterpret_cast<const f4 *>(A.hw + (size_t)x * A.hstride);
#pragma unroll
            for (int i = 0; i < NW4; ++i) w[c][i] = wrow[i];
        }
    };

    // the slots past the 256 a wave fills are read by the last columns' padded taps: zero weights, so any finite value will do
    if (lane < KC_DOWN2_SLOTS - 256u) T[wave][256u + lane] = f4{ 0.0f, 0.0f, 0.0f, 0.0f };

    // ---- vertical pass ----
    f2 alo[4], ahi[4];  // sums of the quad's columns 0, 1 and 2, 3 for the four rows (pairs: one packed operation each)
#pragma unroll
    for (int k = 0; k < 4; ++k) alo[k] = ahi[k] = f2{ 0.0f, 0.0f };
    const uint32_t first = rec0[0], last = rec0[3];  // the group's windows span source rows first .. last; loads are clamped to `last`
    // N rows from `first + row0` on
    auto fetch = [&](auto &p, uint32_t row0) {
        constexpr int N = (int)(sizeof(p) / sizeof(f4));
#pragma unroll
        for (int u = 0; u < N; ++u)
            p[u] = reinterpret_cast<const f4 *>(src0 + (size_t)min(first + row0 + (uint32_t)u, last) * spitch)[qi];
    };
    // ... into the four sums: the N x 4 weights at w (w[4 u + k]: source row u, output row k) and their presence bits at m
    auto consume = [&](const auto &p, d2_const_u32 w, d2_const_u32 m) {
        constexpr int N = (int)(sizeof(p) / sizeof(f4));
        // the weights: N / 4 scalar loads in flight together (left to itself the compiler fetches them one after the other into
        // the same registers, each with a wait of its own, between the arithmetic)
        u16v W[N / 4];
#pragma unroll
        for (int i = 0; i < N / 4; ++i) W[i] = *reinterpret_cast<d2_const_u16v>(w + 16 * i);
        if constexpr (N == 16) asm volatile("" ::"s"(W[0]), "s"(W[1]), "s"(W[2]), "s"(W[3]));
        else asm volatile("" ::"s"(W[0]), "s"(W[1]));
        // A tap that output row k does not have carries the weight +0.0 in the record.  Its product is +-0 and adding that
        // to a sum that started at +0.0 never changes it (such a sum is never -0.0) -- as long as the sample is finite.
        // One test finds the waves for which that fails; they take the exact form below.
        float big = 0.0f;
#pragma unroll
        for (int u = 0; u < N; ++u) {
            big = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_fabsf(p[u].x), __builtin_fabsf(p[u].y)), big);
            big = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_fabsf(p[u].z), __builtin_fabsf(p[u].w)), big);
        }
        if (__builtin_amdgcn_ballot_w64(!(big < __builtin_inff())) == 0ull) {
#pragma unroll
            for (int u = 0; u < N; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t wbits = W[u >> 2][4 * (u & 3) + k];  // (a copy: __builtin_bit_cast of a vector ELEMENT reads element 0)
                    const float wt = __builtin_bit_cast(float, wbits);
                    alo[k] += f2{ p[u].x, p[u].y } * wt;
                    ahi[k] += f2{ p[u].z, p[u].w } * wt;
                }
            }
        } else {
            // some sample is infinite or NaN: absent taps are left out by a per-lane select (the presence mask -- bit 4 u + k:
            // output row k has a tap on source row u -- is the same in every lane; as a vector value it keeps this arm free of
            // branches)
            uint32_t mlo = m[0], mhi = N == 16 ? m[1] : 0u;
            asm volatile("" : "+v"(mlo), "+v"(mhi));
#pragma unroll
            for (int u = 0; u < N; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t wbits = W[u >> 2][4 * (u & 3) + k];  // (a copy: __builtin_bit_cast of a vector ELEMENT reads element 0)
                    const float wt = __builtin_bit_cast(float, wbits);
                    const int bit = 4 * u + k;
                    const bool on = ((bit < 32 ? mlo >> bit : mhi >> (bit - 32)) & 1u) != 0u;
                    const f2 tlo = f2{ p[u].x, p[u].y } * wt, thi = f2{ p[u].z, p[u].w } * wt;
                    alo[k] += on ? tlo : f2{ 0.0f, 0.0f };
                    ahi[k] += on ? thi : f2{ 0.0f, 0.0f };
                }
        }
    };
    // half h of the records: rows 8 h .. 8 h + 7 of the group's span, weights and presence bits inside record h / 2
    auto half_w = [&](uint32_t h) { return rec0 + (h >> 1) * KC_DOWN2_REC + 8u + 32u * (h & 1u); };
    auto half_m = [&](uint32_t h) { return rec0 + (h >> 1) * KC_DOWN2_REC + 1u + (h & 1u); };
    if constexpr (MODE == 0) {  // one chunk: its 16 loads are in flight together, the arithmetic follows them as they arrive
        f4 p[16];
        fetch(p, 0u);
        consume(p, rec0 + 8, rec0 + 1);
    } else {
        // several chunks, eight rows at a time: the next eight are in flight during the arithmetic on these (16 loads in
        // flight as above, but a wave's life is one memory latency plus the arithmetic instead of one latency per chunk; 91
        // registers instead of 105).  Against chunk after chunk: 14.5 / 15.2 us on Gaussian 3000^2 -> 700^2, 24.4 / 26.0 on
        // Gaussian 4096^2 -> 2048^2, 22.2 / 21.2 on CatmullRom 4:1 (profiles/r03_down2_pipe.txt).  Two whole chunks in registers
        // (169 of them, 2 waves per SIMD) lost 15-20 % everywhere (r03_down2_pf.txt): this kernel lives on waves in flight.
        const uint32_t halves = rec0[5];
        f4 pa[8], pb[8];
        fetch(pa, 0u);
        for (uint32_t h = 0; h < halves; h += 2u) {
            if (h + 1u < halves) fetch(pb, 8u * (h + 1u));
            consume(pa, half_w(h), half_m(h));
            if (h + 1u < halves) {
                if (h + 2u < halves) fetch(pa, 8u * (h + 2u));
                consume(pb, half_w(h + 1u), half_m(h + 1u));
            }
        }
    }
    columns();

    // ---- transposition through the wave's own LDS ----
    f4 *Tw = T[wave];
    float big = 0.0f;  // the horizontal pass pads its windows with zero weights too: are the intermediate values finite?
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        big = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_fabsf(alo[k].x), __builtin_fabsf(alo[k].y)), big);
        big = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_fabsf(ahi[k].x), __builtin_fabsf(ahi[k].y)), big);
    }
    const bool finite = __builtin_amdgcn_ballot_w64(!(big < __builtin_inff())) == 0ull;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        Tw[4u * lane + c] = f4{ alo[0][c], alo[1][c], alo[2][c], alo[3][c] };
        Tw[4u * lane + 2 + c] = f4{ ahi[0][c], ahi[1][c], ahi[2][c], ahi[3][c] };
    }
    // the slots are this wave's own: its lanes' writes only have to be ordered before its lanes' reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- horizontal pass ----
    float *dst0 = P.dst[z] + (size_t)(4u * g) * P.dpitch[z] + x0;  // (uniform)
    const uint32_t rows = min(4u, A.dh - 4u * g);
    f4 sum[HC];
    if (finite) {
        // Every lane runs all 4 NW4 taps: the table's rows are padded with +0.0 weights, and +-0 added to a sum that began at
        // +0.0 leaves it as it is.
#pragma unroll
        for (int c = 0; c < HC; ++c) {
            const f4 *t = Tw + h0[c];
            sum[c] = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int j = 0; j < 4 * NW4; ++j) sum[c] += t[j] * w[c][j >> 2][j & 3];
        }
    } else {
        // an infinite or NaN intermediate value somewhere in the wave's rows: taps a column does not have are left out
#pragma unroll
        for (int c = 0; c < HC; ++c) {
            const f4 *t = Tw + h0[c];
            sum[c] = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int j = 0; j < 4 * NW4; ++j) {
                const f4 term = t[j] * w[c][j >> 2][j & 3];
                sum[c] += (uint32_t)j < n[c] ? term : f4{ 0.0f, 0.0f, 0.0f, 0.0f };
            }
        }
    }
#pragma unroll
    for (int c = 0; c < HC; ++c) {
        if (x0 + lane + 64u * c < x1) {
            const f4 v = d2_clamp(sum[c]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((uint32_t)k < rows) (dst0 + (size_t)k * P.dpitch[z] + 64u * c)[lane] = v[k];
        }
    }
}

hipError_t launch_resize_down2(const ResizePlanes &p, int batch, const Down2Args &a, hipStream_t s)
{
    if (a.dw == 0 || a.dh == 0) return hipSuccess;
    if (batch < 1 || batch > 4) return hipErrorInvalidValue;
    const uint32_t nw4 = a.hstride / 4u;
    if (a.hstride % 4u != 0 || nw4 < 1 || nw4 > 8 || a.nc < 1 || a.nc > KC_DOWN2_MAX_CHUNKS || a.tile_w == 0 || !a.strips)
        return hipErrorInvalidValue;
    const uint32_t hc = down2_cols_per_lane(nw4);
    if (a.tile_w > 64u * hc) return hipErrorInvalidValue;
    dim3 grid((a.dw + a.tile_w - 1) / a.tile_w, (a.dh + 15u) / 16u, batch);
    Down2Args a2 = a;
    a2.by_rows = 0;
    // KC_DOWN2_XCD=0 / 1: never / always (A/B); default: when the planes fit the Infinity Cache (a.xcd_per as the caller's hint).
    // Measured (profiles/r03_down2_xcd.txt): one 4096^2 plane 27.9 -> 26.2 us, 3000^2 -> 700^2 17.9 -> 15.4; four 4096^2 planes
    // (268 MB of source, past the cache) 114.6 -> 122.7: there the plain order, whole rows at a time, is kinder to HBM.
    static const int xcd_env = std::getenv("KC_DOWN2_XCD") ? std::atoi(std::getenv("KC_DOWN2_XCD")) : -1;
    const bool xcd = xcd_env < 0 ? a.xcd_per != 0 : xcd_env != 0;
    a2.xcd_per = 0;
    // Four strips of one row group per workgroup and the jobs dealt to the XCDs in eighths row by row (what resize_poly_kernel
    // gained 8 - 20 % from): here it pays where the row groups' windows span several chunks (ratios from about 1.6: CatmullRom
    // 4096^2 -> 1365^2 21.3 -> 19.3 us, RGBA 74.3 -> 63.9; Gaussian 3000^2 -> 700^2 14.6 -> 12.5, RGBA 52.4 -> 41.4; Lanczos3 2:1
    // 24.8 -> 23.4) and costs 5 - 10 % on RGBA launches of single-chunk ratios (4096^2 -> 3000^2 111.5 -> 122.4), which keep the
    // order above (profiles/r04_down2_by_rows.txt).
    if (a.by_rows && grid.x >= 2) {  // (the caller's choice: resize.cpp, kc_set_option("down2_by_rows"); one strip: nothing to order,
                                     // and the reciprocal of 1 does not fit 32 bits)
        const uint32_t gx = grid.x, groups = (a.dh + 3u) / 4u;
        const uint64_t jobs = (uint64_t)gx * groups, magic = ((1ull << 32) + gx - 1) / gx;
        if (jobs < (1u << 24) && jobs * (magic * gx - (1ull << 32)) < (1ull << 32)) {
            a2.by_rows = 1u;
            a2.n_tiles = (uint32_t)jobs;
            a2.gy = gx;
            a2.gy_magic = (uint32_t)magic;
            a2.xcd_per = (uint32_t)(((jobs + 3u) / 4u + 7u) / 8u);
            grid = dim3(8u * a2.xcd_per, 1, batch);
        }
    } else
    if (xcd && grid.y >= 2) {
        const uint64_t n = (uint64_t)grid.x * grid.y, magic = ((1ull << 32) + grid.y - 1) / grid.y;
        // tile / gy == (tile * magic) >> 32 for every tile < n when n * (magic * gy - 2^32) < 2^32
        if (n < (1u << 24) && n * (magic * grid.y - (1ull << 32)) < (1ull << 32)) {
            a2.n_tiles = (uint32_t)n;
            a2.gy = grid.y;
            a2.gy_magic = (uint32_t)magic;
            a2.xcd_per = (uint32_t)((n + 7) / 8);
            grid = dim3(8u * a2.xcd_per, 1, batch);
        }
    }
#define KC_D2(HC, NW4)                                                                     \
    do {                                                                                   \
        if (a.nc == 1) resize_down2_kernel<HC, NW4, 0><<<grid, 256, 0, s>>>(p, a2);        \
        else resize_down2_kernel<HC, NW4, 1><<<grid, 256, 0, s>>>(p, a2);                  \
    } while (0)
    switch (nw4) {
    case 1: KC_D2(3, 1); break;
    case 2: KC_D2(3, 2); break;
    case 3: KC_D2(3, 3); break;
    case 4: KC_D2(2, 4); break;
    case 5: KC_D2(1, 5); break;
    case 6: KC_D2(1, 6); break;
    case 7: KC_D2(1, 7); break;
    default: KC_D2(1, 8); break;
    }
#undef KC_D2
    return hipGetLastError();
}

}  // namespace kc
