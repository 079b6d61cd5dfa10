// Hand-written CDNA4 (gfx950) kernels for the kanter_core per-pixel hot path.
//
// All of these are HBM-bandwidth-bound pointwise / small-stencil kernels: 16-byte (dwordx4)
// coalesced row-major accesses on 256-byte-pitched f32 planes, 64-wide wavefronts, no MFMA.
// Build flags that matter for parity with the reference's scalar Rust loops:
//   -ffp-contract=off                         no FMA contraction (Rust never fuses a*b+c)
//   -fhip-fp32-correctly-rounded-divide-sqrt  IEEE f32 divide / sqrt
//   f32 denormals are not flushed (gfx9 default)
// Reference paths are relative to the reference checkout.
#include "kc_internal.hpp"

#include <algorithm>
#include <cstdlib>

namespace kc {

// Grid cap of the grid-stride streaming kernels (to_u8, from_u8, height_to_normal); KC_TUNE_CAP overrides (tuning).
// Default: no cap, one quad / pixel per thread -- from_u8 58.1 -> 50.4 us, height_to_normal 56.8 -> 55.4 us at 4096^2
// against 8192 workgroups looping twice (profiles/r02_kernel_times.txt); to_u8 does not care.
static uint64_t grid_cap(uint64_t dflt)
{
    static long v = [] {
        const char *e = std::getenv("KC_TUNE_CAP");
        return e ? std::atol(e) : 0L;
    }();
    return v > 0 ? (uint64_t)v : dflt;
}

#include "chain_apply.inc"  // splat4, f4, kc_powf, apply1<CODE>: shared with chain1.hip

// Loads / stores with the launch's cache policy (ChainProgram::nt_mask; runtime.cpp, cache_policy_mask) as a compile-time
// property: NT = the stream does not fit the Infinity Cache and is marked nontemporal.
template <bool NT, class V>
static __device__ __forceinline__ V ld_policy(const V *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, class V>
static __device__ __forceinline__ void st_policy(V *p, V v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// One step on the U float4 a thread owns, written OUT OF PLACE (dst = op(src, x)): the decode
// loop ping-pongs between two register sets, so no switch arm ever has to preserve or merge the
// old accumulator and the step costs exactly one packed VALU instruction per pixel pair.
template <int CODE, int U>
static __device__ __forceinline__ void apply4(f4 (&dst)[U], const f4 (&src)[U], const f4 (&x)[U], float c, const PowCtx *tab)
{
#pragma unroll
    for (int u = 0; u < U; ++u) {
        dst[u].x = apply1<CODE>(src[u].x, x[u].x, c, tab);
        dst[u].y = apply1<CODE>(src[u].y, x[u].y, c, tab);
        dst[u].z = apply1<CODE>(src[u].z, x[u].z, c, tab);
        dst[u].w = apply1<CODE>(src[u].w, x[u].w, c, tab);
    }
}

template <int CODE, int U>
static __device__ __forceinline__ void apply4c(f4 (&dst)[U], const f4 (&src)[U], float c, const PowCtx *tab)
{
#pragma unroll
    for (int u = 0; u < U; ++u) {
        dst[u].x = apply1<CODE>(src[u].x, c, 0.0f, tab);
        dst[u].y = apply1<CODE>(src[u].y, c, 0.0f, tab);
        dst[u].z = apply1<CODE>(src[u].z, c, 0.0f, tab);
        dst[u].w = apply1<CODE>(src[u].w, c, 0.0f, tab);
    }
}

// Decode.  The step record {word, constant} is wave-uniform (kernel argument block, read through
// SMEM one step ahead), so the dispatch is scalar compares and branches and each arm is
// straight-line VALU on the U float4 the thread owns.  PMC shows the kernel is ISSUE-bound for
// long chains (every instruction, scalar or vector, costs the wave ~4 issue cycles), so the
// scalar path is kept short: one 16-byte scalar load per two steps, a 2-level switch (operand
// source, then op), and only the arms the program can contain -- MODE 0 = {+, -, *}, 1 = + divide,
// 2 = + pow (f64 pow call).  x + acc / x * acc are canonicalised to acc + x / acc * x on the host.
#define KC_CODE_SWITCH(APPLY, DST, SRC)                                                   \
    switch (w & 0xffu) {                                                                  \
    case CH_ADD: APPLY(CH_ADD, DST, SRC); break;                                          \
    case CH_SUB_L: APPLY(CH_SUB_L, DST, SRC); break;                                      \
    case CH_SUB_R: APPLY(CH_SUB_R, DST, SRC); break;                                      \
    case CH_MUL: APPLY(CH_MUL, DST, SRC); break;                                          \
    case CH_DIV_L: if constexpr (MODE >= 1) { APPLY(CH_DIV_L, DST, SRC); } else __builtin_unreachable(); break; \
    case CH_DIV_R: if constexpr (MODE >= 1) { APPLY(CH_DIV_R, DST, SRC); } else __builtin_unreachable(); break; \
    case CH_POW_L: if constexpr (MODE >= 2) { APPLY(CH_POW_L, DST, SRC); } else __builtin_unreachable(); break; \
    case CH_POW_R: if constexpr (MODE >= 2) { APPLY(CH_POW_R, DST, SRC); } else __builtin_unreachable(); break; \
    default: __builtin_unreachable();                                                     \
    }

// Plane operands also carry the fused "step, then c - acc" codes.
#define KC_CODE_SWITCH_P(APPLY, DST, SRC)                                                 \
    switch (w & 0xffu) {                                                                  \
    case CH_ADD: APPLY(CH_ADD, DST, SRC); break;                                          \
    case CH_SUB_L: APPLY(CH_SUB_L, DST, SRC); break;                                      \
    case CH_SUB_R: APPLY(CH_SUB_R, DST, SRC); break;                                      \
    case CH_MUL: APPLY(CH_MUL, DST, SRC); break;                                          \
    case CH_ADD_INV: APPLY(CH_ADD_INV, DST, SRC); break;                                  \
    case CH_SUBL_INV: APPLY(CH_SUBL_INV, DST, SRC); break;                                \
    case CH_SUBR_INV: APPLY(CH_SUBR_INV, DST, SRC); break;                                \
    case CH_MUL_INV: APPLY(CH_MUL_INV, DST, SRC); break;                                  \
    case CH_DIV_L: if constexpr (MODE >= 1) { APPLY(CH_DIV_L, DST, SRC); } else __builtin_unreachable(); break; \
    case CH_DIV_R: if constexpr (MODE >= 1) { APPLY(CH_DIV_R, DST, SRC); } else __builtin_unreachable(); break; \
    case CH_POW_L: if constexpr (MODE >= 2) { APPLY(CH_POW_L, DST, SRC); } else __builtin_unreachable(); break; \
    case CH_POW_R: if constexpr (MODE >= 2) { APPLY(CH_POW_R, DST, SRC); } else __builtin_unreachable(); break; \
    default: __builtin_unreachable();                                                     \
    }

// Runs the whole step program on the U float4 a thread holds: acc = start, then every step.
template <int K, int U, int MODE>
static __device__ __forceinline__ void chain_run(const ChainProgram &P, const uint32_t b, const f4 (&in)[K][U], f4 (&acc)[U],
                                                 const PowCtx *tab = nullptr)
{
    if (P.start_src < 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = f4{ P.start_c[b], P.start_c[b], P.start_c[b], P.start_c[b] };
    } else {
        switch (P.start_src) {
        case 0:
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = in[0][u];
            break;
        case 1:
            if constexpr (K > 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u] = in[1][u];
            }
            break;
        case 2:
            if constexpr (K > 2) {
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u] = in[2][u];
            }
            break;
        default:
            if constexpr (K > 3) {
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u] = in[3][u];
            }
            break;
        }
    }

    // Steps alternate acc -> alt -> acc (no arm ever merges register sets); the host validates
    // every record, so unknown words cannot occur.
    const uint32_t n_ops = P.n_ops;
    const ChainStepPair *pp = P.step[b];  // two records per 16-byte scalar load, fetched one pair ahead
    ChainStepPair nxt = pp[0];
    f4 alt[U];
#define KC_APPLY_C(CODE, DST, SRC) apply4c<CODE, U>(DST, SRC, c, tab)
#define KC_APPLY_0(CODE, DST, SRC) apply4<CODE, U>(DST, SRC, in[0], c, tab)
#define KC_APPLY_1(CODE, DST, SRC) apply4<CODE, U>(DST, SRC, in[K > 1 ? 1 : 0], c, tab)
#define KC_APPLY_2(CODE, DST, SRC) apply4<CODE, U>(DST, SRC, in[K > 2 ? 2 : 0], c, tab)
#define KC_APPLY_3(CODE, DST, SRC) apply4<CODE, U>(DST, SRC, in[K > 3 ? 3 : 0], c, tab)
#define KC_STEP(DST, SRC, REC)                                                          \
    {                                                                                   \
        const uint32_t w = (REC).word;                                                  \
        const float c = (REC).c;                                                        \
        switch (w >> 8) {                                                               \
        case 0: KC_CODE_SWITCH(KC_APPLY_C, DST, SRC) break;                             \
        case 1: KC_CODE_SWITCH_P(KC_APPLY_0, DST, SRC) break;                           \
        case 2: if constexpr (K > 1) { KC_CODE_SWITCH_P(KC_APPLY_1, DST, SRC) } else __builtin_unreachable(); break; \
        case 3: if constexpr (K > 2) { KC_CODE_SWITCH_P(KC_APPLY_2, DST, SRC) } else __builtin_unreachable(); break; \
        case 4: if constexpr (K > 3) { KC_CODE_SWITCH_P(KC_APPLY_3, DST, SRC) } else __builtin_unreachable(); break; \
        default: __builtin_unreachable();                                               \
        }                                                                               \
    }
    uint32_t i = 0;
    for (; i + 1 < n_ops; i += 2) {
        const ChainStepPair cur = nxt;
        nxt = pp[i / 2 + 1];
        KC_STEP(alt, acc, cur.a)
        KC_STEP(acc, alt, cur.b)
    }
    if (i < n_ops) {
        KC_STEP(alt, acc, nxt.a)
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = alt[u];
    }
#undef KC_STEP
#undef KC_APPLY_C
#undef KC_APPLY_0
#undef KC_APPLY_1
#undef KC_APPLY_2
#undef KC_APPLY_3
}

// Fused Mix chain (src/node/mix.rs:136-192 applied N times without materialising the
// intermediates).  K = distinct input planes, U = float4 per thread per decode, MODE = op set.
// Algorithmic HBM bytes per pixel: 4 * (planes read + 1 written), whatever N is.
// NT: the launch's cache policy marks streams (ChainProgram::nt_mask != 0): every full-size input is read and the result stored
// with the nontemporal hint.  (The interpreter runs a program's first two sightings only; it does not distinguish which input
// the policy would have kept cacheable -- the kernels compiled for the program do.)
template <int K, int U, int MODE, bool NT = false>
__global__ __launch_bounds__(256) void chain_kernel(const ChainProgram P)
{
    __shared__ double pow_lds[MODE >= 2 ? KC_POW_TABLE_DOUBLES : 1];
    PowCtx pw{};
    if constexpr (MODE >= 2) pw = pow_setup(pow_lds);
    const PowCtx *pow_tab = &pw;
    const uint32_t b = blockIdx.y;
    const uint32_t total = P.rows * P.row_units;
    const bool flat = P.rows == 1;
    const f4 *inp[K];
    uint32_t ipitch[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        inp[k] = reinterpret_cast<const f4 *>(P.in[b][k]);
        ipitch[k] = P.in_pitch[b][k];
    }
    f4 *outp = reinterpret_cast<f4 *>(P.out[b]);
    const uint32_t opitch = P.out_pitch[b];
    const uint32_t step = gridDim.x * (256u * U);

    for (uint32_t base = blockIdx.x * (256u * U) + threadIdx.x; base < total; base += step) {
        f4 in[K][U];
        f4 acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t idx = base + u * 256u;
            uint32_t row = 0, col = idx;
            if (!flat) {
                row = idx / P.row_units;
                col = idx - row * P.row_units;
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
                in[k][u] = idx < total ? ld_policy<NT>(&inp[k][row * ipitch[k] + col]) : f4{ 0.0f, 0.0f, 0.0f, 0.0f };
        }

        chain_run<K, U, MODE>(P, b, in, acc, pow_tab);

        // output offsets are recomputed here rather than kept live across the program (VGPRs)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t idx = base + u * 256u;
            uint32_t row = 0, col = idx;
            if (!flat) {
                row = idx / P.row_units;
                col = idx - row * P.row_units;
            }
            if (idx < total) st_policy<NT>(&outp[row * opitch + col], acc[u]);
        }
    }
}

// Zero-input chain (constant start, constant operands only): still one pass of stores.
template <int MODE>
__global__ __launch_bounds__(256) void chain_kernel_k0(const ChainProgram P)
{
    __shared__ double pow_lds[MODE >= 2 ? KC_POW_TABLE_DOUBLES : 1];
    PowCtx pw{};
    if constexpr (MODE >= 2) pw = pow_setup(pow_lds);
    const PowCtx *tab = &pw;
    const uint32_t b = blockIdx.y;
    const uint32_t total = P.rows * P.row_units;
    const bool flat = P.rows == 1;
    f4 *outp = reinterpret_cast<f4 *>(P.out[b]);
    const uint32_t opitch = P.out_pitch[b];
    f4 acc[1];
    acc[0] = f4{ P.start_c[b], P.start_c[b], P.start_c[b], P.start_c[b] };
    for (uint32_t i = 0; i < P.n_ops; ++i) {
        const ChainStepRec r = (i & 1u) ? P.step[b][i / 2].b : P.step[b][i / 2].a;
        const uint32_t w = r.word;
        const float c = r.c;
        f4 nxt[1];
#define KC_APPLY_C(CODE, DST, SRC) apply4c<CODE, 1>(DST, SRC, c, tab)
        KC_CODE_SWITCH(KC_APPLY_C, nxt, acc)
#undef KC_APPLY_C
        acc[0] = nxt[0];
    }
    const uint32_t step = gridDim.x * 256u;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += step) {
        uint32_t row = 0, col = idx;
        if (!flat) {
            row = idx / P.row_units;
            col = idx - row * P.row_units;
        }
        outp[row * opitch + col] = acc[0];
    }
}

template <int U, int MODE>
static hipError_t launch_chain_k(const ChainProgram &p, dim3 grid, hipStream_t s)
{
    // the nontemporal form exists for the default shapes only (U = 4 without pow, U = 1 with): tuning overrides stay plain
    constexpr bool HAS_NT = (MODE < 2 && U == 4) || (MODE == 2 && U == 1);
    if constexpr (HAS_NT) {
        if (p.nt_mask != 0) {
            switch (p.n_in) {
            case 1: chain_kernel<1, U, MODE, true><<<grid, 256, 0, s>>>(p); return hipGetLastError();
            case 2: chain_kernel<2, U, MODE, true><<<grid, 256, 0, s>>>(p); return hipGetLastError();
            case 3: chain_kernel<3, U, MODE, true><<<grid, 256, 0, s>>>(p); return hipGetLastError();
            case 4: chain_kernel<4, U, MODE, true><<<grid, 256, 0, s>>>(p); return hipGetLastError();
            default: break;
            }
        }
    }
    switch (p.n_in) {
    case 0: chain_kernel_k0<MODE><<<grid, 256, 0, s>>>(p); break;
    case 1: chain_kernel<1, U, MODE><<<grid, 256, 0, s>>>(p); break;
    case 2: chain_kernel<2, U, MODE><<<grid, 256, 0, s>>>(p); break;
    case 3: chain_kernel<3, U, MODE><<<grid, 256, 0, s>>>(p); break;
    case 4: chain_kernel<4, U, MODE><<<grid, 256, 0, s>>>(p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int U, int MODE>
static hipError_t launch_chain_u(const ChainProgram &p, int batch, uint64_t total, int max_blocks, hipStream_t s)
{
    uint64_t blocks = (total + 256 * U - 1) / (256 * U);
    if (blocks > (uint64_t)max_blocks) blocks = max_blocks;
    return launch_chain_k<U, MODE>(p, dim3((unsigned)blocks, batch, 1), s);
}

hipError_t launch_chain(const ChainProgram &p, int batch, int mode, int max_blocks, int unroll, hipStream_t s)
{
    if (batch < 1 || batch > KC_CHAIN_MAX_BATCH || p.n_ops > KC_CHAIN_MAX_OPS || p.n_ops < 1) return hipErrorInvalidValue;
    const uint64_t total = (uint64_t)p.rows * p.row_units;
    if (total == 0) return hipSuccess;
    if (total > 0xFFFFFFFFull) return hipErrorInvalidValue;
    if (mode >= 2) return launch_chain_u<1, 2>(p, batch, total, max_blocks, s);
    if (mode == 1) return launch_chain_u<4, 1>(p, batch, total, max_blocks, s);
    // U = float4 per lane per decode.  U = 4 (74-106 VGPRs, 4-6 waves/SIMD) is the measured optimum
    // on MI355X for 1-64 step chains: U = 2 doubles the scalar decode work per pixel, U = 8 drops to
    // 2-3 waves/SIMD (profiles/r01_chain_unroll.md).  KC_CHAIN_UNROLL overrides for tuning.
    switch (unroll) {
    case 1: return launch_chain_u<1, 0>(p, batch, total, max_blocks, s);
    case 2: return launch_chain_u<2, 0>(p, batch, total, max_blocks, s);
    case 6: return launch_chain_u<6, 0>(p, batch, total, max_blocks, s);
    case 8: return launch_chain_u<8, 0>(p, batch, total, max_blocks, s);
    default: return launch_chain_u<4, 0>(p, batch, total, max_blocks, s);
    }
}

// vec![v; n] (src/slot_image.rs:28-64): only when a constant plane must really exist in HBM.
__global__ __launch_bounds__(256) void fill_kernel(float4 *dst, uint32_t pitch4, uint32_t row_units, uint32_t rows,
                                                   float v)
{
    const uint32_t total = rows * row_units;
    const float4 val = splat4(v);
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const uint32_t row = idx / row_units;
        const uint32_t col = idx - row * row_units;
        dst[row * pitch4 + col] = val;
    }
}

hipError_t launch_fill(float *dst, uint32_t pitch_floats, uint32_t w, uint32_t h, float v, hipStream_t s)
{
    const uint32_t row_units = (w + 3) / 4;
    const uint64_t total = (uint64_t)row_units * h;
    if (total == 0) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    // (a 64 MiB plane: 13.0 / 12.0 / 11.2 us with 1024 / 4096 / 16384 workgroups -- profiles/r04_write_bench.txt; 11.2 us = 6.0 TB/s
    // is what the memory system takes as writes from any kernel shape tried there)
    if (blocks > 16384) blocks = 16384;
    fill_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(reinterpret_cast<float4 *>(dst), pitch_floats / 4, row_units, h,
                                                        v);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Separable resample = image::imageops::resize (crate image 0.24.0) as called from
// src/shared.rs:159-199.  Weights come from host-built tap tables (resize.cpp) so they are the
// same f32 values the scalar algorithm computes; the sums run sequentially from 0.0, unfused.
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ float clamp01_nan_passthrough(float t)
{
    // image::math::utils::clamp: NaN compares false both ways and passes through.
    if (t < 0.0f) return 0.0f;
    if (t > 1.0f) return 1.0f;
    return t;
}

// Pass 1 of the two-pass form: tmp[oy][x] = sum_j src[left_v[oy] + j][x] * w_v[oy][j].
__global__ __launch_bounds__(256) void resize_vertical_kernel(const float *__restrict__ src, uint32_t spitch,
                                                              uint32_t sw, float *__restrict__ tmp, uint32_t tpitch,
                                                              TapsDev V)
{
    const uint32_t oy = blockIdx.y;
    const uint32_t left = V.left[oy];
    const uint32_t n = V.count[oy];
    const float *w = V.w + (size_t)oy * V.stride;
    for (uint32_t x = blockIdx.x * 256u + threadIdx.x; x < sw; x += gridDim.x * 256u) {
        float t = 0.0f;
        for (uint32_t j = 0; j < n; ++j) t += src[(size_t)(left + j) * spitch + x] * w[j];
        tmp[(size_t)oy * tpitch + x] = t;
    }
}

// Pass 2: dst[oy][ox] = clamp(sum_j tmp[oy][left_h[ox] + j] * w_h[ox][j], 0, 1).
__global__ __launch_bounds__(256) void resize_horizontal_kernel(const float *__restrict__ tmp, uint32_t tpitch,
                                                                float *__restrict__ dst, uint32_t dpitch,
                                                                uint32_t dw, TapsDev H)
{
    const uint32_t oy = blockIdx.y;
    for (uint32_t ox = blockIdx.x * 256u + threadIdx.x; ox < dw; ox += gridDim.x * 256u) {
        const uint32_t left = H.left[ox];
        const uint32_t n = H.count[ox];
        const float *w = H.w + (size_t)ox * H.stride;
        float t = 0.0f;
        for (uint32_t j = 0; j < n; ++j) t += tmp[(size_t)oy * tpitch + left + j] * w[j];
        dst[(size_t)oy * dpitch + ox] = clamp01_nan_passthrough(t);
    }
}

hipError_t launch_resize_vertical(const float *src, uint32_t spitch, uint32_t sw, float *tmp, uint32_t tpitch,
                                  uint32_t dh, TapsDev v, hipStream_t s)
{
    if (sw == 0 || dh == 0) return hipSuccess;
    uint32_t bx = (sw + 255) / 256;
    if (bx > 64) bx = 64;
    resize_vertical_kernel<<<dim3(bx, dh), 256, 0, s>>>(src, spitch, sw, tmp, tpitch, v);
    return hipGetLastError();
}

hipError_t launch_resize_horizontal(const float *tmp, uint32_t tpitch, float *dst, uint32_t dpitch, uint32_t dw,
                                    uint32_t dh, TapsDev h, hipStream_t s)
{
    if (dw == 0 || dh == 0) return hipSuccess;
    uint32_t bx = (dw + 255) / 256;
    if (bx > 64) bx = 64;
    resize_horizontal_kernel<<<dim3(bx, dh), 256, 0, s>>>(tmp, tpitch, dst, dpitch, dw, h);
    return hipGetLastError();
}

// Single-pass tiled form: each workgroup owns a tile_h x tile_w output tile.
//   phase 1 vertical pass HBM -> LDS: (tile row, 4-column group) items dealt out to all lanes; a
//           lane reads its item's source rows with 16-byte loads, several in flight, weights from
//           the LDS copy of the tile rows' tap table (resize_vpass_items).  Rows shared by
//           neighbouring output rows are re-read through L1/L2, not HBM.  The intermediate the
//           two-pass form would write to HBM (tile_h x ncp floats) never leaves the CU;
//   phase 2 horizontal pass out of LDS: every thread owns 4 consecutive output columns for the
//           whole tile, so its tap windows and weights sit in registers and the
//           four results leave as one 16-byte store -- or, in resize_chain_kernel, feed the Mix
//           chain that consumes the resampled plane without ever being written.
// Same operands, same order, same roundings as the two-pass form: bit-identical output.
// Algorithmic bytes per output pixel = 4 * (1 + in_px / out_px).
struct ResizeTile {
    uint32_t x0, y0, x1, y1, th, c0;
    const float *tmp;  // tile_h x ncp vertical-pass intermediate in LDS
};

template <int MAXT>
struct ResizeCols {  // the 4 output columns a thread owns: window start (tile-relative), weights, which taps exist
    typedef float f2 __attribute__((ext_vector_type(2)));
    uint32_t hl[4];
    f2 w01[MAXT], w23[MAXT];  // the weights of columns 0, 1 and 2, 3 as the packed multiply takes them
    bool live[4][MAXT];
    uint32_t minc;  // fewest taps of the four
};

template <int MAXT>
static __device__ __forceinline__ void resize_load_cols(ResizeCols<MAXT> &C, const TapsDev &H, uint32_t ox, uint32_t x1,
                                                        uint32_t c0)
{
    C.minc = 0xFFFFFFFFu;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t x = min(ox + e, x1 - 1);
        C.hl[e] = H.left[x] - c0;
        const uint32_t hn = H.count[x];
        C.minc = min(C.minc, hn);
        const float *wh = H.w + (size_t)x * H.stride;
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
            (e < 2 ? C.w01[j] : C.w23[j])[e & 1] = wh[j];  // rows of the table are zero-padded to `stride` entries
            C.live[e][j] = (uint32_t)j < hn;
        }
    }
}

// Phase 1 for the workgroup's tile; ends with the barrier that publishes `tmp`.
// `src` rows are 16-byte aligned (plane pitch is a multiple of 16 bytes), so the window starts at
// c0 = first source column rounded down to a multiple of 4; the last group may run past the
// source width into the row's pitch padding -- those intermediates are never read by phase 2.
// Work items are (tile row, 4-column group) pairs dealt out to all 256 lanes; a lane walks its
// item's window VU source rows at a time (VU independent 16-byte loads in flight), its taps read
// from the LDS copy of the tile rows' tap table.  Everything is per lane: no scalar-unit work
// beyond the loop counters (the scalar unit is shared by the CU's four SIMDs).
// Taps every row of the tile has (j < vmin) are summed unconditionally; the remaining ones are
// per-lane predicated: a tap past a lane's window repeats its last row and adds -0.0.
// SWZ: the intermediate row is stored with one float of padding after every 32 (index i lives at i + (i >> 5)), the
// layout resize_wide_kernel's horizontal pass reads without bank conflicts; `ncp4` is then unused and `ncp_swz` is the
// row pitch in floats.
template <int VU, bool SWZ = false>
static __device__ __forceinline__ void resize_vpass_items(const f4 *__restrict__ src4, uint32_t sp4, f4 *tmp4, uint32_t ncp4,
                                                          uint32_t th, uint32_t nq, const uint32_t *vl, const uint32_t *vn,
                                                          const float *vw, uint32_t vstride, uint32_t vmin, uint32_t ncp_swz = 0)
{
    auto add = [](f4 &a, const f4 &px, float wt) {
        a.x += px.x * wt;
        a.y += px.y * wt;
        a.z += px.z * wt;
        a.w += px.w * wt;
    };
    const uint32_t items = th * nq;
    // i / nq by multiply-high: exact here because i < th * nq <= 4096 (16 bytes of LDS per item, 64 KiB)
    const uint32_t nq_magic = nq > 1 ? 0xFFFFFFFFu / nq + 1u : 0u;
    for (uint32_t i = threadIdx.x; i < items; i += 256u) {
        const uint32_t ty = nq > 1 ? __umulhi(i, nq_magic) : i;
        const uint32_t q = i - ty * nq;
        const uint32_t n = vn[ty];
        const f4 *col = src4 + (size_t)vl[ty] * sp4 + q;
        const float *w = vw + ty * vstride;
        f4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
        uint32_t j0 = 0;
        for (; j0 + VU <= vmin; j0 += VU) {
            f4 p[VU];
            float wt[VU];
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                p[u] = col[(size_t)(j0 + u) * sp4];
                wt[u] = w[j0 + u];
            }
#pragma unroll
            for (int u = 0; u < VU; ++u) add(acc, p[u], wt[u]);
        }
        if constexpr (VU > 4) {
            // (windows of 5 .. 7 taps everywhere in the tile -- up-sampling with CatmullRom, Lanczos3, Gaussian: four of them need
            // no predicate either)
            for (; j0 + 4u <= vmin; j0 += 4u) {
                f4 p[4];
                float wt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    p[u] = col[(size_t)(j0 + u) * sp4];
                    wt[u] = w[j0 + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) add(acc, p[u], wt[u]);
            }
        }
        for (; j0 < vstride; j0 += 4u) {
            f4 p[4];
            float wt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                p[u] = col[(size_t)min(j0 + u, n - 1u) * sp4];
                wt[u] = w[min(j0 + u, vstride - 1u)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool live = j0 + u < n;
                acc.x += live ? p[u].x * wt[u] : -0.0f;
                acc.y += live ? p[u].y * wt[u] : -0.0f;
                acc.z += live ? p[u].z * wt[u] : -0.0f;
                acc.w += live ? p[u].w * wt[u] : -0.0f;
            }
        }
        if constexpr (SWZ) {
            float *o = reinterpret_cast<float *>(tmp4) + ty * ncp_swz + 4u * q + (q >> 3);  // a quad never straddles a multiple of 32
            o[0] = acc.x;
            o[1] = acc.y;
            o[2] = acc.z;
            o[3] = acc.w;
        } else {
            tmp4[ty * ncp4 + q] = acc;
        }
    }
}

// SWZ (resize_wide_kernel): rows of `tmp` are ncp + ncp / 32 + 1 floats apart and swizzled, see resize_vpass_items.
template <bool SWZ = false>
static __device__ __forceinline__ ResizeTile resize_tile_vpass(float *lds, const float *__restrict__ src,
                                                               uint32_t spitch, uint32_t dw, uint32_t dh,
                                                               const TapsDev &V, const TapsDev &H, uint32_t tile_w,
                                                               uint32_t tile_h, uint32_t ncp)
{
    const uint32_t row_floats = SWZ ? ncp + (ncp >> 5) + 1u : ncp;
    ResizeTile T;
    T.x0 = blockIdx.x * tile_w;
    T.y0 = blockIdx.y * tile_h;
    T.x1 = min(T.x0 + tile_w, dw);
    T.y1 = min(T.y0 + tile_h, dh);
    T.th = T.y1 - T.y0;
    T.c0 = H.left[T.x0] & ~3u;
    T.tmp = lds;
    const uint32_t nq = (H.left[T.x1 - 1] + H.count[T.x1 - 1] - T.c0 + 3u) / 4u;  // <= ncp / 4 (host-checked)

    // the tile rows' vertical taps: one coalesced fetch into LDS
    uint32_t *vl = reinterpret_cast<uint32_t *>(lds + tile_h * row_floats + 8u);
    uint32_t *vn = vl + tile_h;
    float *vw = reinterpret_cast<float *>(vn + tile_h);  // tile_h x V.stride
    for (uint32_t i = threadIdx.x; i < T.th; i += 256u) {
        vl[i] = V.left[T.y0 + i];
        vn[i] = V.count[T.y0 + i];
    }
    for (uint32_t i = threadIdx.x; i < T.th * V.stride; i += 256u) vw[i] = V.w[(size_t)T.y0 * V.stride + i];
    __syncthreads();
    // fewest taps of any row of this tile (tile_h <= 64: one value per lane, butterfly minimum)
    uint32_t vmin = (threadIdx.x & 63u) < T.th ? vn[threadIdx.x & 63u] : 0xFFFFFFFFu;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmin = min(vmin, (uint32_t)__shfl_xor((int)vmin, off));
    vmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)vmin);

    const f4 *src4 = reinterpret_cast<const f4 *>(src + T.c0);
    f4 *tmp4 = reinterpret_cast<f4 *>(lds);
    if (V.stride <= 4u)
        resize_vpass_items<4, SWZ>(src4, spitch / 4u, tmp4, ncp / 4u, T.th, nq, vl, vn, vw, V.stride, vmin, row_floats);
    else
        resize_vpass_items<8, SWZ>(src4, spitch / 4u, tmp4, ncp / 4u, T.th, nq, vl, vn, vw, V.stride, vmin, row_floats);
    __syncthreads();
    return T;
}

// Horizontal pass for one tile row and this thread's 4 columns.  Taps j < MINT need no predicate (every lane of the wave has
// them); columns 0, 1 and 2, 3 go through the packed multiply and add as pairs (their weights sit in register pairs for the
// whole tile; as four separate sums the compiler packed products of one column's neighbouring taps and then shuffled them
// apart again for the adds: 35 moves per row and thread).
template <int MINT, int MAXT>
static __device__ __forceinline__ void resize_out_row(const ResizeCols<MAXT> &C, const float *row, float (&res)[4])
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    // Taps are contiguous from hl[e]: one base address per output, constant offsets per tap.  A tap past the window reads the
    // next floats of the LDS block -- always inside the allocation (the vertical tap table follows tmp) -- and is discarded below.
    f2 t01 = { 0.0f, 0.0f }, t23 = { 0.0f, 0.0f };
    const float *pe[4] = { row + C.hl[0], row + C.hl[1], row + C.hl[2], row + C.hl[3] };
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
        f2 q01 = f2{ pe[0][j], pe[1][j] } * C.w01[j];
        f2 q23 = f2{ pe[2][j], pe[3][j] } * C.w23[j];
        if (j >= MINT) {
            // a tap that does not exist contributes -0.0: t + (-0.0) == t for every t (including +-0, +-inf, NaN), so the sum
            // equals the reference's shorter sum
            q01.x = C.live[0][j] ? q01.x : -0.0f;
            q01.y = C.live[1][j] ? q01.y : -0.0f;
            q23.x = C.live[2][j] ? q23.x : -0.0f;
            q23.y = C.live[3][j] ? q23.y : -0.0f;
        }
        t01 += q01;
        t23 += q23;
    }
    res[0] = clamp01_nan_passthrough(t01.x);
    res[1] = clamp01_nan_passthrough(t01.y);
    res[2] = clamp01_nan_passthrough(t23.x);
    res[3] = clamp01_nan_passthrough(t23.y);
}

// How many taps a wave may sum without a predicate: all of its lanes' columns have MAXT - 2 (windows of 6 or 8 register taps:
// up-sampling by a non-integer ratio has 4 - 5 taps with CatmullRom, 6 - 7 with Lanczos3 / Gaussian) or MAXT - 1 (4 register
// taps), else what the whole image guarantees.
template <int MINT, int MAXT>
struct ResizeUmin {
    static constexpr int value = MAXT >= 6 ? MAXT - 2 : MAXT == 4 ? 3 : MINT;
};

template <int MINT, int MAXT>  // horizontal taps, all in registers: MINT unconditional, up to MAXT
__global__ __launch_bounds__(256) void resize_lds_kernel(const ResizePlanes P, uint32_t dw, uint32_t dh, TapsDev V,
                                                         TapsDev H, uint32_t tile_w, uint32_t tile_h, uint32_t ncp)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float *__restrict__ src = P.src[blockIdx.z];  // blockIdx.z = plane: up to 4 planes of one image per launch
    float *__restrict__ dst = P.dst[blockIdx.z];
    const uint32_t spitch = P.spitch[blockIdx.z], dpitch = P.dpitch[blockIdx.z];
    const uint32_t col_groups = tile_w / 4;         // threads across one tile row
    const uint32_t row_groups = 256u / col_groups;  // tile rows in flight
    const uint32_t cg = threadIdx.x % col_groups;
    const uint32_t rg = threadIdx.x / col_groups;
    const uint32_t x0 = blockIdx.x * tile_w, x1 = min(x0 + tile_w, dw);
    const uint32_t ox = x0 + 4 * cg;
    // this thread's 4 output columns (fetched first so the loads overlap the staging)
    ResizeCols<MAXT> C;
    resize_load_cols<MAXT>(C, H, ox, x1, H.left[x0] & ~3u);
    const ResizeTile T = resize_tile_vpass(lds, src, spitch, dw, dh, V, H, tile_w, tile_h, ncp);
    constexpr int UMIN = ResizeUmin<MINT, MAXT>::value;
    // (asked of every lane, also those without columns: their minc is that of the tile's last column)
    const bool wave_has_umin = UMIN > MINT && __builtin_amdgcn_ballot_w64(C.minc < (uint32_t)UMIN) == 0ull;
    if (ox >= T.x1) return;
    if (ox + 3 < T.x1) {
        // interior columns: one 16-byte store per row
        if (wave_has_umin) {
            for (uint32_t ty = rg; ty < T.th; ty += row_groups) {
                float res[4];
                resize_out_row<UMIN, MAXT>(C, T.tmp + ty * ncp, res);
                *reinterpret_cast<float4 *>(dst + (size_t)(T.y0 + ty) * dpitch + ox) = make_float4(res[0], res[1], res[2], res[3]);
            }
        } else {
            for (uint32_t ty = rg; ty < T.th; ty += row_groups) {
                float res[4];
                resize_out_row<MINT, MAXT>(C, T.tmp + ty * ncp, res);
                *reinterpret_cast<float4 *>(dst + (size_t)(T.y0 + ty) * dpitch + ox) = make_float4(res[0], res[1], res[2], res[3]);
            }
        }
    } else {
        // the tile's last, partial quad
        for (uint32_t ty = rg; ty < T.th; ty += row_groups) {
            float res[4];
            resize_out_row<MINT, MAXT>(C, T.tmp + ty * ncp, res);
            float *o = dst + (size_t)(T.y0 + ty) * dpitch + ox;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ox + e < T.x1) o[e] = res[e];
        }
    }
}

// Wide horizontal windows (more than 8 taps: down-sampling).  The tile's horizontal tap table is
// staged in LDS behind the vertical one and every thread produces single outputs, four taps per trip
// (both operands come from LDS; a tap past the window repeats the last one and adds -0.0).
__global__ __launch_bounds__(256) void resize_wide_kernel(const ResizePlanes P, uint32_t dw, uint32_t dh, TapsDev V,
                                                          TapsDev H, uint32_t tile_w, uint32_t tile_h, uint32_t ncp,
                                                          uint32_t h_off)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float *__restrict__ src = P.src[blockIdx.z];
    float *__restrict__ dst = P.dst[blockIdx.z];
    const uint32_t spitch = P.spitch[blockIdx.z], dpitch = P.dpitch[blockIdx.z];
    const uint32_t x0 = blockIdx.x * tile_w, tw = min(x0 + tile_w, dw) - x0;
    const uint32_t c0 = H.left[x0] & ~3u;
    uint32_t *hl = reinterpret_cast<uint32_t *>(lds + h_off);
    uint32_t *hn = hl + tile_w;
    float *hw = reinterpret_cast<float *>(hn + tile_w);  // tile_w x H.stride
    for (uint32_t i = threadIdx.x; i < tw; i += 256u) {
        hl[i] = H.left[x0 + i] - c0;
        hn[i] = H.count[x0 + i];
    }
    for (uint32_t i = threadIdx.x; i < tw * H.stride; i += 256u) hw[i] = H.w[(size_t)x0 * H.stride + i];
    const ResizeTile T = resize_tile_vpass<true>(lds, src, spitch, dw, dh, V, H, tile_w, tile_h, ncp);  // its barriers publish hl/hn/hw
    const uint32_t row_floats = ncp + (ncp >> 5) + 1u;
    const uint32_t sh = 31u - (uint32_t)__clz((int)tile_w);  // tile_w is a power of two
    for (uint32_t i = threadIdx.x; i < T.th * tile_w; i += 256u) {
        const uint32_t ty = i >> sh, x = i & (tile_w - 1u);
        if (x >= tw) continue;
        const uint32_t n = hn[x];
        // Neighbouring outputs read windows `ratio` floats apart: straight indexing put the 32 lanes of a pass on 8 (ratio 4)
        // or 4 (ratio 8) banks -- PMC: 65 % of this pass's LDS cycles were bank conflicts.  The intermediate row is stored
        // with one float of padding after every 32 (resize_vpass_items<.., true>), which spreads strides 2, 4 and 8 over all banks.
        const float *row = T.tmp + ty * row_floats;
        const uint32_t h0 = hl[x];
        const float *w = hw + x * H.stride;
        float t = 0.0f;
        for (uint32_t j0 = 0; j0 < n; j0 += 4u) {
            float p[4], wt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t jj = min(j0 + u, n - 1u);
                const uint32_t idx = h0 + jj;
                p[u] = row[idx + (idx >> 5)];
                wt[u] = w[jj];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) t += j0 + u < n ? p[u] * wt[u] : -0.0f;
        }
        dst[(size_t)(T.y0 + ty) * dpitch + x0 + x] = clamp01_nan_passthrough(t);
    }
}

// Down-sampling (more than 8 taps on BOTH axes), second form.  resize_wide_kernel's vertical pass gathers every output
// row's whole window from global memory (25 16-byte loads per row and column quad for Lanczos3 4:1) with per-lane tap
// look-ups; here the unit of work is wave-uniform instead:
//   vertical pass   a WAVE owns R adjacent tile rows and 64 column quads.  It walks the union of the R windows once,
//                   four source rows per trip (one 16-byte load per lane and row), and feeds each row into every sum
//                   whose window contains it.  Which sums those are depends only on the rows, not on the lane, so the
//                   tests are scalar branches on scalar-loaded window bounds and the weights arrive as scalar loads from
//                   the tap table: (taps + (R - 1) ratio) / R loads per output row instead of taps, no tap staging, no
//                   per-lane control.  Each sum still receives its taps in ascending order: same roundings.
//   horizontal pass a lane owns one output column for four tile rows at a time: one weight read and one swizzled
//                   index per tap serve four sums.
// Rows whose windows are far apart (the wrapped rows of a row band) fall back to one row per walk.
// What bounds it (profiles/r02_down_kernel.md): the windows of neighbouring row groups overlap, the overlap is re-read
// by another wave several trips later and by then has left L2 -- 88 % of this pass's reads miss it -- so the pass runs at
// the fabric's rate on (taps + (R - 1) ratio) / (R ratio) times the plane.
template <int R>
static __device__ __forceinline__ void resize_down_rows(const f4 *__restrict__ src4, uint32_t sp4, float *tmp,
                                                        uint32_t row_floats, uint32_t nq, uint32_t ty0, uint32_t th,
                                                        uint32_t y0, const TapsDev &V, uint32_t lane)
{
    uint32_t left[R], cnt[R];
    const float *w[R];
    uint32_t smin = 0xFFFFFFFFu, smax = 0u;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool on = ty0 + k < th;
        const uint32_t y = y0 + (on ? ty0 + k : ty0);
        left[k] = V.left[y];
        cnt[k] = on ? V.count[y] : 0u;
        w[k] = V.w + (size_t)y * V.stride;
        if (on) {
            smin = min(smin, left[k]);
            smax = max(smax, left[k] + cnt[k]);
        }
    }
    for (uint32_t qb = 0; qb < nq; qb += 64u) {
        const uint32_t q = min(qb + lane, nq - 1u);
        const f4 *col = src4 + q;
        f4 acc[R];
#pragma unroll
        for (int k = 0; k < R; ++k) acc[k] = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
        for (uint32_t s0 = smin; s0 < smax; s0 += 4u) {
            f4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = col[(size_t)min(s0 + u, smax - 1u) * sp4];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const uint32_t j0 = s0 - left[k];  // wraps for rows above the window
                if (s0 >= left[k] && j0 + 3u < cnt[k]) {
                    const float w0 = w[k][j0], w1 = w[k][j0 + 1u], w2 = w[k][j0 + 2u], w3 = w[k][j0 + 3u];
                    acc[k] += p[0] * w0;
                    acc[k] += p[1] * w1;
                    acc[k] += p[2] * w2;
                    acc[k] += p[3] * w3;
                } else if (s0 + 3u >= left[k] && s0 < left[k] + cnt[k]) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t j = j0 + (uint32_t)u;
                        if (j < cnt[k]) acc[k] += p[u] * w[k][j];  // j wraps to a huge value above the window
                    }
                }
            }
        }
        if (qb + lane < nq) {
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (ty0 + k < th) {
                    float *o = tmp + (ty0 + k) * row_floats + 4u * q + (q >> 3);  // a quad never straddles a multiple of 32
                    o[0] = acc[k].x;
                    o[1] = acc[k].y;
                    o[2] = acc[k].z;
                    o[3] = acc[k].w;
                }
        }
    }
}

// The strip's horizontal taps, staged once per workgroup (every wave of it works on the same output columns).
struct DownStrip {
    uint32_t x0, tw, c0, nq, row_floats, hsp;
    float *tmp;
    uint32_t *hl, *hn;
    float *hw;
};

static __device__ __forceinline__ DownStrip resize_down_stage(float *lds, const TapsDev &H, uint32_t dw, uint32_t tile_w,
                                                              uint32_t tmp_rows, uint32_t ncp, uint32_t bx)
{
    DownStrip S;
    S.x0 = bx * tile_w;
    const uint32_t x1 = min(S.x0 + tile_w, dw);
    S.tw = x1 - S.x0;
    S.c0 = H.left[S.x0] & ~3u;
    S.nq = (H.left[x1 - 1] + H.count[x1 - 1] - S.c0 + 3u) / 4u;  // <= ncp / 4 (host-checked)
    S.row_floats = KC_DOWN_ROW_FLOATS;  // a constant: the horizontal pass addresses its four rows with immediate offsets
    S.hsp = H.stride | 1u;  // odd pitch: the lanes' weight rows start on different banks
    S.tmp = lds;
    S.hl = reinterpret_cast<uint32_t *>(lds + tmp_rows * S.row_floats);
    S.hn = S.hl + tile_w;
    S.hw = reinterpret_cast<float *>(S.hn + tile_w);  // tile_w x hsp
    for (uint32_t i = threadIdx.x; i < S.tw; i += 256u) {
        S.hl[i] = H.left[S.x0 + i] - S.c0;
        S.hn[i] = H.count[S.x0 + i];
    }
    for (uint32_t i = threadIdx.x; i < S.tw * H.stride; i += 256u) {
        const uint32_t x = i / H.stride, j = i - x * H.stride;
        S.hw[x * S.hsp + j] = H.w[(size_t)S.x0 * H.stride + i];
    }
    return S;
}

// The same for ONE wave (resize_poly_kernel's band waves, each with a strip of its own): `lds` is the wave's own area -- four
// intermediate rows, then the taps -- filled by its 64 lanes; the caller orders it with a wave barrier.
static __device__ __forceinline__ DownStrip resize_down_stage_wave(float *lds, const TapsDev &H, uint32_t dw, uint32_t tile_w, uint32_t bx,
                                                                   uint32_t lane)
{
    DownStrip S;
    S.x0 = bx * tile_w;
    const uint32_t x1 = min(S.x0 + tile_w, dw);
    S.tw = x1 - S.x0;
    S.c0 = H.left[S.x0] & ~3u;
    S.nq = (H.left[x1 - 1] + H.count[x1 - 1] - S.c0 + 3u) / 4u;  // <= 64 (host-checked)
    S.row_floats = KC_DOWN_ROW_FLOATS;
    S.hsp = H.stride | 1u;
    S.tmp = lds;
    S.hl = reinterpret_cast<uint32_t *>(lds + 4u * S.row_floats);
    S.hn = S.hl + tile_w;
    S.hw = reinterpret_cast<float *>(S.hn + tile_w);  // tile_w x hsp
    for (uint32_t i = lane; i < S.tw; i += 64u) {
        S.hl[i] = H.left[S.x0 + i] - S.c0;
        S.hn[i] = H.count[S.x0 + i];
    }
    for (uint32_t i = lane; i < S.tw * H.stride; i += 64u) {
        const uint32_t x = i / H.stride, j = i - x * H.stride;
        S.hw[x * S.hsp + j] = H.w[(size_t)S.x0 * H.stride + i];
    }
    return S;
}
// floats of such an area
static inline uint32_t resize_down_wave_floats(uint32_t tile_w, uint32_t hstride) { return (4u * KC_DOWN_ROW_FLOATS + 2u * tile_w + tile_w * (hstride | 1u) + 3u) / 4u * 4u; }

// Horizontal pass of four intermediate rows (row, row + row_floats, ...) for this lane's output column.
static __device__ __forceinline__ void resize_down_hrows(const DownStrip &S, const float *row, uint32_t lane, float *dst_row,
                                                         uint32_t dpitch, uint32_t nrows)
{
    const uint32_t n = S.hn[lane], h0 = S.hl[lane];
    const float *w = S.hw + lane * S.hsp;
    // rows 0, 1 and rows 2, 3 as pairs: one packed multiply and add per pair and tap
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 t01 = { 0.0f, 0.0f }, t23 = { 0.0f, 0.0f };
    // Every column of the strip has the same number of taps, a multiple of 4 (the interior of an integer-ratio resample):
    // no tap needs clamping or masking.
    const uint32_t nu = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    if ((nu & 3u) == 0u && __builtin_amdgcn_ballot_w64(n != nu) == 0ull) {
        for (uint32_t j0 = 0; j0 < nu; j0 += 4u) {
            f2 p01[4], p23[4];
            float wt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t idx = h0 + j0 + u;
                const float *v = row + idx + (idx >> 5);
                wt[u] = w[j0 + u];
                p01[u] = f2{ v[0], v[KC_DOWN_ROW_FLOATS] };
                p23[u] = f2{ v[2 * KC_DOWN_ROW_FLOATS], v[3 * KC_DOWN_ROW_FLOATS] };
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                t01 += p01[u] * wt[u];
                t23 += p23[u] * wt[u];
            }
        }
    } else {
        for (uint32_t j0 = 0; j0 < n; j0 += 4u) {
            f2 p01[4], p23[4];
            float wt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t jj = min(j0 + u, n - 1u);
                const uint32_t idx = h0 + jj;
                const float *v = row + idx + (idx >> 5);
                wt[u] = w[jj];
                p01[u] = f2{ v[0], v[KC_DOWN_ROW_FLOATS] };
                p23[u] = f2{ v[2 * KC_DOWN_ROW_FLOATS], v[3 * KC_DOWN_ROW_FLOATS] };
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // a tap that does not exist contributes -0.0, which leaves every sum unchanged
                const bool live = j0 + u < n;
                t01 += live ? p01[u] * wt[u] : f2{ -0.0f, -0.0f };
                t23 += live ? p23[u] * wt[u] : f2{ -0.0f, -0.0f };
            }
        }
    }
    const float t[4] = { t01.x, t01.y, t23.x, t23.y };
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if ((uint32_t)r < nrows) dst_row[(size_t)r * dpitch + S.x0 + lane] = clamp01_nan_passthrough(t[r]);
}

// One tile of 4 R rows at y0 (th of them exist): the general form, any windows.  Called by all four waves.
template <int R>
static __device__ __forceinline__ void resize_down_tile(const DownStrip &S, const float *__restrict__ src, uint32_t spitch,
                                                        float *__restrict__ dst, uint32_t dpitch, uint32_t y0, uint32_t th,
                                                        const TapsDev &V, uint32_t wave, uint32_t lane)
{
    const f4 *src4 = reinterpret_cast<const f4 *>(src + S.c0);
    {
        const uint32_t ty0 = wave * R;
        // the union of the wave's windows, against the windows themselves
        uint32_t lo = 0xFFFFFFFFu, hi = 0u, sum = 0u;
        for (uint32_t k = 0; k < R && ty0 + k < th; ++k) {
            const uint32_t l = V.left[y0 + ty0 + k], n = V.count[y0 + ty0 + k];
            lo = min(lo, l);
            hi = max(hi, l + n);
            sum += n;
        }
        if (hi - lo <= sum) {
            resize_down_rows<R>(src4, spitch / 4u, S.tmp, S.row_floats, S.nq, ty0, th, y0, V, lane);
        } else {
            for (uint32_t k = 0; k < R && ty0 + k < th; ++k)
                resize_down_rows<1>(src4, spitch / 4u, S.tmp, S.row_floats, S.nq, ty0 + k, th, y0, V, lane);
        }
    }
    __syncthreads();
    if (lane < S.tw)
        for (uint32_t tb = wave * 4u; tb < th; tb += 16u)
            resize_down_hrows(S, S.tmp + tb * S.row_floats, lane, dst + (size_t)(y0 + tb) * dpitch, dpitch, th - tb);
}

template <int R>  // tile rows per wave: the tile is 4 R rows high
__global__ __launch_bounds__(256) void resize_down_kernel(const ResizePlanes P, uint32_t dw, uint32_t dh, TapsDev V,
                                                          TapsDev H, uint32_t tile_w, uint32_t ncp)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr uint32_t tile_h = 4u * R;
    const DownStrip S = resize_down_stage(lds, H, dw, tile_w, tile_h, ncp, blockIdx.x);
    const uint32_t y0 = blockIdx.y * tile_h, th = min(y0 + tile_h, dh) - y0;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    resize_down_tile<R>(S, P.src[blockIdx.z], P.spitch[blockIdx.z], P.dst[blockIdx.z], P.dpitch[blockIdx.z], y0, th, V, wave,
                        threadIdx.x & 63u);
}

// Integer-ratio down-sampling (4096 -> 1024, -> 512, -> 2048 ...): away from the image border every output row has the
// same n = A * RT weights and its window starts RT source rows below its neighbour's (the host checks this bit for bit,
// TapsHost::reg_*).  A wave then STREAMS a band of 12 such rows: RT source rows per trip; the sum of "age" a (the row
// whose window began a trips ago) receives its taps a RT .. a RT + RT - 1 from them; after the trip the oldest sum is
// complete, goes to a four-row LDS ring and the sums move up one age.  No window test, no weight fetch (the A * RT weights
// sit in scalar registers), every source row of the band is loaded once, and after each four finished rows the wave runs the
// horizontal pass on its ring by itself: no barrier after the tap staging.  Same taps in the same order: same roundings.
// Rows near the border (and what does not fill a band) are tiles of the general form, run by the launch's last workgroups.
// Which band wave sits where (round 4): the four waves of a workgroup are four NEIGHBOURING STRIPS of one band (their windows
// share halo columns: one L1), each with its strip's taps staged in LDS by itself, and workgroup id % 8 -- the XCD -- works
// through the k-th eighth of the bands one whole band after the other (a band and the next one share A - 1 trips of rows: one
// L2; every XCD streams a contiguous eighth of the plane).  The bare access pattern takes 12.9 us like this against 18.8 with
// four bands of one strip per workgroup (profiles/tile_read_bench.hip), the kernels 8 - 20 % less.
struct PolyBands {
    uint32_t ya, yb;    // regular rows handled as bands: [ya, yb), yb - ya a multiple of 4
    uint32_t rows;      // rows per band (a multiple of 4; the last band may be shorter)
    uint32_t n_bands;
    uint32_t ty0[4], th[4];  // general tiles: first row, rows (<= 16)
    // A workgroup's four waves are four neighbouring STRIPS of one band (each with its taps staged by itself); the grid is
    // one-dimensional and XCD k (workgroup id % 8) works through the k-th eighth of the band workgroups band by band; the
    // border tiles follow.  n_sq strip quads per band, n_band_wgs = n_bands * n_sq, xper = ceil(n_band_wgs / 8), gx strips,
    // wave_floats of LDS per wave.
    uint32_t n_sq, n_band_wgs, xper, gx, wave_floats;
};


template <int A, int RT>
__global__ __launch_bounds__(256) void resize_poly_kernel(const ResizePlanes P, uint32_t dw, uint32_t dh, TapsDev V,
                                                          TapsDev H, uint32_t tile_w, uint32_t ncp, PolyBands B)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float *__restrict__ src = P.src[blockIdx.z];
    float *__restrict__ dst = P.dst[blockIdx.z];
    const uint32_t spitch = P.spitch[blockIdx.z], dpitch = P.dpitch[blockIdx.z];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sp4 = spitch / 4u;
    DownStrip S;
    uint32_t yf;
    f4 pn[RT];
    const f4 *col;
    // ---- four strips of one band per workgroup, the bands dealt to the XCDs in eighths (profiles/r04_poly_weights.md 5a: the
    // halo columns of neighbouring strips meet in one L1, a band and the next one in one L2) ----
    if (blockIdx.x >= 8u * B.xper) {
        const uint32_t g = blockIdx.x - 8u * B.xper, t = g / B.gx;
        S = resize_down_stage(lds, H, dw, tile_w, 16u, ncp, g - t * B.gx);
        resize_down_tile<4>(S, src, spitch, dst, dpitch, B.ty0[t], B.th[t], V, wave, lane);
        return;
    }
    const uint32_t tile = (blockIdx.x & 7u) * B.xper + (blockIdx.x >> 3);
    if (tile >= B.n_band_wgs) return;
    const uint32_t bi = tile / B.n_sq, strip = (tile - bi * B.n_sq) * 4u + wave;
    if (strip >= B.gx) return;  // (no workgroup barrier on this path)
    yf = B.ya + B.rows * bi;
    // the band's first rows are requested before the taps are staged: both are in flight together
    {
        const uint32_t x0 = strip * tile_w, x1 = min(x0 + tile_w, dw), c0 = H.left[x0] & ~3u;
        const uint32_t nq = (H.left[x1 - 1] + H.count[x1 - 1] - c0 + 3u) / 4u;
        col = reinterpret_cast<const f4 *>(src + c0) + min(lane, nq - 1u) + (size_t)V.left[yf] * sp4;
    }
#pragma unroll
    for (int u = 0; u < RT; ++u) pn[u] = col[(size_t)u * sp4];
    S = resize_down_stage_wave(lds + wave * B.wave_floats, H, dw, tile_w, strip, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t ROWS = min(B.rows, B.yb - yf);
    const bool q_ok = lane < S.nq;
    const uint32_t q = min(lane, S.nq - 1u);
    // The A RT weights, two to a VECTOR register pair (the same values in every lane); either half of a pair is broadcast to both
    // lanes of a packed multiply by op_sel, at no cost.  As scalar registers -- rounds 2 and 3 -- the compiler wanted every
    // weight as an SGPR PAIR (w, w) for v_pk_mul_f32: 96 scalar registers at ratio 8, which do not exist, so it parked them in
    // the lanes of two vector registers and fetched each pair back with two v_readlane and a wait state before its multiply:
    // 112 of the 304 vector instructions of a trip (208 now; Gaussian 8:1 30.4 -> 27.7 us, Lanczos3 4:1 22.2 -> 21.1:
    // profiles/r04_poly_weights.md).
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 Wp[A][RT / 2];
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int u = 0; u < RT; u += 2)
            Wp[a][u / 2] = f2{ V.w[(size_t)B.ya * V.stride + a * RT + u], V.w[(size_t)B.ya * V.stride + a * RT + u + 1] };
    auto mad = [&](f4 &sum, const f4 &p, int a, int u) {
        // (the empty statement keeps the pair where it is and the broadcast inside the loop: hoisted out of it, the 48 splats
        // would be 96 more registers -- tried: 303 VGPRs, slower)
        asm volatile("" : "+v"(Wp[a][u / 2]));
        const f2 wp = Wp[a][u / 2];
        const f2 w2 = (u & 1) ? __builtin_shufflevector(wp, wp, 1, 1) : __builtin_shufflevector(wp, wp, 0, 0);
        const f2 lo = f2{ p.x, p.y } * w2, hi = f2{ p.z, p.w } * w2;
        sum = f4{ sum.x + lo.x, sum.y + lo.y, sum.z + hi.x, sum.w + hi.y };
    };
    float *ring = S.tmp;
    float *ringq = ring + 4u * q + (q >> 3);  // swizzled: a quad never straddles a multiple of 32
    f4 acc[A];
#pragma unroll
    for (int a = 0; a < A; ++a) acc[a] = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
    const uint32_t TRIPS = ROWS + A - 1;
    for (uint32_t c = 0; c < TRIPS; ++c) {
        f4 p[RT];
#pragma unroll
        for (int u = 0; u < RT; ++u) p[u] = pn[u];
        if (c + 1u < TRIPS) {
#pragma unroll
            for (int u = 0; u < RT; ++u) pn[u] = col[(size_t)((c + 1u) * RT + u) * sp4];
        }
        // age a holds row c - a of the band
        if (c >= (uint32_t)(A - 1) && c < ROWS) {
#pragma unroll
            for (int u = 0; u < RT; ++u)
#pragma unroll
                for (int a = 0; a < A; ++a) mad(acc[a], p[u], a, u);
        } else {
#pragma unroll
            for (int a = 0; a < A; ++a)
                if (c >= (uint32_t)a && c - (uint32_t)a < ROWS) {
#pragma unroll
                    for (int u = 0; u < RT; ++u) mad(acc[a], p[u], a, u);
                }
        }
        if (c >= (uint32_t)(A - 1)) {
            const uint32_t k = c - (uint32_t)(A - 1);  // this row of the band is complete
            if (q_ok) {
                float *o = ringq + (k & 3u) * S.row_floats;
                o[0] = acc[A - 1].x;
                o[1] = acc[A - 1].y;
                o[2] = acc[A - 1].z;
                o[3] = acc[A - 1].w;
            }
            if ((k & 3u) == 3u) {
                // the ring is this wave's own: its lanes' writes only have to be ordered before its lanes' reads
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < S.tw) resize_down_hrows(S, ring, lane, dst + (size_t)(yf + k - 3u) * dpitch, dpitch, 4u);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
#pragma unroll
        for (int a = A - 1; a > 0; --a) acc[a] = acc[a - 1];
        acc[0] = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
    }
}

// Fused resample + Mix chain: phase 2's four results are input slot K-1 of the chain program, the
// other K-1 inputs are resident planes read with one 16-byte load each, and only the chain's
// result is stored.  The resampled plane itself never exists in HBM: per output pixel the launch
// moves 4 * (K - 1 + 1) bytes plus the (small) source tile instead of 4 * (1 + K + 1).
// blockIdx.z = channel (each channel resamples its own source plane with the shared tap tables).
template <int K, int MAXT>
__global__ __launch_bounds__(256) void resize_chain_kernel(const ChainProgram P, uint32_t dw, uint32_t dh, TapsDev V,
                                                           TapsDev H, uint32_t tile_w, uint32_t tile_h, uint32_t ncp)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t b = blockIdx.z;
    const uint32_t col_groups = tile_w / 4;
    const uint32_t row_groups = 256u / col_groups;
    const uint32_t cg = threadIdx.x % col_groups;
    const uint32_t rg = threadIdx.x / col_groups;
    const uint32_t x0 = blockIdx.x * tile_w, x1 = min(x0 + tile_w, dw);
    const uint32_t ox = x0 + 4 * cg;
    const uint32_t y0 = blockIdx.y * tile_h;
    const uint32_t th = min(tile_h, dh - y0);
    constexpr int KM = K > 1 ? K - 1 : 1;
    const f4 *inp[KM];
    uint32_t ipitch[KM];
#pragma unroll
    for (int k = 0; k < K - 1; ++k) {
        inp[k] = reinterpret_cast<const f4 *>(P.in[b][k]) + ox / 4;  // the whole quad lies inside the pitch
        ipitch[k] = P.in_pitch[b][k];
    }
    // RU tile rows per trip: the chain program is decoded once for RU float4 (its scalar decode is
    // the expensive part, see chain_run); rows past the tile repeat its last row and are not stored.
    // With one resident input, its quads for trip i + 1 are requested before trip i is computed (the
    // first before the vertical pass): a wave then never waits for loads queued behind its own stores.
    constexpr int RU = 4;  // 2 rows per trip: 89.4 us, 1 row: 95.6 us, 4 rows: 83.5 us on config #2 (profiles/r02_fused_ru_ab.txt)
    constexpr bool AHEAD = K <= 2;  // 16 more registers per resident input: not worth the occupancy beyond one
    f4 nxt[KM][RU];
    auto request = [&](uint32_t ty0) {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const uint32_t oy = y0 + min(ty0 + u * row_groups, th - 1);
#pragma unroll
            for (int k = 0; k < K - 1; ++k) nxt[k][u] = inp[k][oy * ipitch[k]];
        }
    };
    if (AHEAD && ox < x1) request(rg);
    ResizeCols<MAXT> C;
    resize_load_cols<MAXT>(C, H, ox, x1, H.left[x0] & ~3u);
    const ResizeTile T = resize_tile_vpass(lds, P.samp_src[b], P.samp_pitch[b], dw, dh, V, H, tile_w, tile_h, ncp);
    if (ox >= T.x1) return;
    float *outp = P.out[b];
    const uint32_t opitch = P.out_pitch[b] * 4;  // floats
    const bool full = ox + 3 < T.x1;
    for (uint32_t ty0 = rg; ty0 < T.th; ty0 += RU * row_groups) {
        f4 in[K][RU];
        if (!AHEAD) request(ty0);
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int k = 0; k < K - 1; ++k) in[k][u] = nxt[k][u];
        if (AHEAD && ty0 + RU * row_groups < T.th) request(ty0 + RU * row_groups);
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const uint32_t ty = min(ty0 + u * row_groups, T.th - 1);
            float res[4];
            resize_out_row<1, MAXT>(C, T.tmp + ty * ncp, res);
            in[K - 1][u] = f4{ res[0], res[1], res[2], res[3] };
        }
        f4 acc[RU];
        chain_run<K, RU, 0>(P, b, in, acc);
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const uint32_t ty = ty0 + u * row_groups;
            if (ty >= T.th) break;
            float *o = outp + (size_t)(T.y0 + ty) * opitch + ox;
            if (full) {
                *reinterpret_cast<f4 *>(o) = acc[u];
            } else {
                const float r4[4] = { acc[u].x, acc[u].y, acc[u].z, acc[u].w };
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ox + e < T.x1) o[e] = r4[e];
            }
        }
    }
}

template <int MINT>
static void launch_resize_lds_t(dim3 grid, size_t lds, hipStream_t s, uint32_t maxt, const ResizePlanes &p, uint32_t dw,
                                uint32_t dh, TapsDev v, TapsDev h, uint32_t tile_w, uint32_t tile_h, uint32_t ncp)
{
#define KC_RESIZE_LAUNCH(MAXT) \
    resize_lds_kernel<(MINT <= MAXT ? MINT : MAXT), MAXT><<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, tile_h, ncp)
    if (maxt <= 1) KC_RESIZE_LAUNCH(1);
    else if (maxt == 2) KC_RESIZE_LAUNCH(2);
    else if (maxt == 3) KC_RESIZE_LAUNCH(3);
    else if (maxt == 4) KC_RESIZE_LAUNCH(4);
    else if (maxt <= 6) KC_RESIZE_LAUNCH(6);
    else KC_RESIZE_LAUNCH(8);
#undef KC_RESIZE_LAUNCH
}

hipError_t launch_resize_lds(const ResizePlanes &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h,
                             uint32_t h_min_count, uint32_t tile_w, uint32_t tile_h, uint32_t ncp, hipStream_t s)
{
    if (dw == 0 || dh == 0) return hipSuccess;
    if (batch < 1 || batch > 4) return hipErrorInvalidValue;
    if (tile_w % 4 != 0 || tile_w > 1024 || 256u % (tile_w / 4) != 0 || tile_h > 64) return hipErrorInvalidValue;
    const size_t lds = resize_lds_bytes(tile_h, ncp, v.stride, tile_w, h.stride);
    dim3 grid((dw + tile_w - 1) / tile_w, (dh + tile_h - 1) / tile_h, batch);
    if (h.stride > KC_RESIZE_REG_TAPS)
        resize_wide_kernel<<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, tile_h, ncp,
                                                  (uint32_t)(lds / sizeof(float) - (2u * tile_w + (size_t)tile_w * h.stride)));
    else if (h_min_count >= 2)
        launch_resize_lds_t<2>(grid, lds, s, h.stride, p, dw, dh, v, h, tile_w, tile_h, ncp);
    else
        launch_resize_lds_t<1>(grid, lds, s, h.stride, p, dw, dh, v, h, tile_w, tile_h, ncp);
    return hipGetLastError();
}

hipError_t launch_resize_down(const ResizePlanes &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h, uint32_t tile_w,
                              uint32_t tile_h, uint32_t ncp, hipStream_t s)
{
    if (dw == 0 || dh == 0) return hipSuccess;
    if (batch < 1 || batch > 4) return hipErrorInvalidValue;
    if (tile_w == 0 || tile_w > 64 || (tile_h != 16 && tile_h != 32) || ncp % 4 != 0 || ncp > 256) return hipErrorInvalidValue;
    const size_t lds = resize_down_lds_bytes(tile_h, ncp, tile_w, h.stride);
    dim3 grid((dw + tile_w - 1) / tile_w, (dh + tile_h - 1) / tile_h, batch);
    if (tile_h == 16)
        resize_down_kernel<4><<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, ncp);
    else
        resize_down_kernel<8><<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, ncp);
    return hipGetLastError();
}

template <int A>
static void launch_resize_poly_a(dim3 grid, size_t lds, hipStream_t s, uint32_t rt, const ResizePlanes &p, uint32_t dw, uint32_t dh,
                                 TapsDev v, TapsDev h, uint32_t tile_w, uint32_t ncp, const PolyBands &b)
{
    if (rt == 2) resize_poly_kernel<A, 2><<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, ncp, b);
    else if (rt == 4) resize_poly_kernel<A, 4><<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, ncp, b);
    else resize_poly_kernel<A, 8><<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, ncp, b);
}

// Rows [reg_a, reg_b) of the vertical table are regular: `ages` x `ratio` taps each, windows `ratio` apart, equal weights.
hipError_t launch_resize_poly(const ResizePlanes &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h, uint32_t tile_w,
                              uint32_t ncp, uint32_t reg_a, uint32_t reg_b, uint32_t ages, uint32_t ratio, hipStream_t s)
{
    if (dw == 0 || dh == 0) return hipSuccess;
    if (batch < 1 || batch > 4) return hipErrorInvalidValue;
    if (tile_w == 0 || tile_w > 64 || ncp % 4 != 0 || ncp > 256 || reg_a > reg_b || reg_b > dh) return hipErrorInvalidValue;
    if ((ages != 2 && ages != 4 && ages != 6) || (ratio != 2 && ratio != 4 && ratio != 8)) return hipErrorInvalidValue;
    PolyBands b{};
    b.ya = reg_a;
    b.yb = reg_a + (reg_b - reg_a) / 4u * 4u;
    // Band height: 12 rows for one plane, 24 for several planes of a long-windowed filter.  Measured on one box
    // (profiles/r02_down_kernel.md): 8 / 12 / 16 rows give 28.7 / 24.5 / 27.7 us on Lanczos3 4:1 -- shorter bands re-read more
    // of their neighbours' windows, taller ones leave too few waves to overlap one wave's arithmetic with another's loads
    // (bands sized for one wave per SIMD, 20+ rows, were slower still); with four planes per launch the waves are there and
    // 12 / 24 / 36 / 48 rows give 80.5 / 71.2 / 74.8 / 85.4 us.
    // A launch whose waves are all resident at once lasts as long as ONE wave lives -- A - 1 + rows trips -- so images that do
    // not fill the chip take SHORT bands: 2048^2 -> 512^2 19.3 / 15.3 / 11.5 us with 12 / 8 / 4 rows, 2048^2 -> 256^2
    // 26.6 / 20.3 / 16.5, 1024^2 -> 256^2 16.7 / 13.2 / 9.6 (profiles/r04_poly_rows_small.txt); past about one wave per SIMD the
    // windows short bands re-read cost more than their trips save (4096^2 -> 1024^2: 22.2 / 24.7 / 33.5 us).
    static const uint32_t rows_env = std::getenv("KC_POLY_ROWS") ? std::max(4u, (uint32_t)std::atoi(std::getenv("KC_POLY_ROWS")) / 4u * 4u) : 0u;
    // With four strips of a band per workgroup (round 4), same run, one plane, 8 / 12 / 16 / 24 rows: Lanczos3 4096^2 -> 1024^2
    // 23.5 / 22.4 / 26.1 / 29.2 us; RGBA launches 12 / 16 / 24 / 32 rows: Lanczos3 4:1 68.1 / 65.8 / 70.8 / 63.5, CatmullRom 4:1
    // 57.3 / 62.7 / 55.7 / 54.2, but 2048^2 -> 512^2 RGBA 26.4 / 26.3 / 34.2 / 34.4 (profiles/r04_poly_rows_by_band.txt): tall bands
    // only where 12-row bands already give six waves per SIMD.
    b.rows = 12u;
    b.gx = (dw + tile_w - 1) / tile_w;
    {
        const uint64_t regular = (reg_b - reg_a) / 4u * 4u;
        bool small = false;
        for (uint32_t r : { 4u, 8u })
            if (b.gx * ((regular + r - 1) / r) * (uint64_t)batch <= 1100u) {
                b.rows = r;
                small = true;
                break;
            }
        if (!small && b.gx * ((regular + 11u) / 12u) * (uint64_t)batch >= 6000u) b.rows = 32u;
    }
    if (rows_env) b.rows = rows_env;
    b.n_bands = (b.yb - b.ya + b.rows - 1) / b.rows;
    // what is left: rows above the first band and below the last one, as general tiles of at most 16 rows
    uint32_t nt = 0;
    auto add_tiles = [&](uint32_t y0, uint32_t y1) {
        for (uint32_t y = y0; y < y1; y += 16u) {
            if (nt == 4) return false;
            b.ty0[nt] = y;
            b.th[nt] = std::min(16u, y1 - y);
            ++nt;
        }
        return true;
    };
    if (!add_tiles(0, b.ya) || !add_tiles(b.yb, dh)) return hipErrorInvalidValue;
    // Band workgroups: four neighbouring strips of one band each, dealt to the XCDs in eighths of the band-major sequence (what
    // this order is worth, same run, us: Gaussian 4096^2 -> 512^2 30.6 -> 25.2, Lanczos3 -> 1024^2 24.7 -> 22.4, Triangle -> 512^2
    // 18.1 -> 16.0, 8192^2 -> 1024^2 83.9 -> 67.8; RGBA Triangle 57.1 -> 48.5 = 0.70 of the HBM peak: profiles/r04_poly_by_band.txt)
    b.n_sq = (b.gx + 3u) / 4u;
    b.n_band_wgs = b.n_bands * b.n_sq;
    b.xper = (b.n_band_wgs + 7u) / 8u;
    b.wave_floats = resize_down_wave_floats(tile_w, h.stride);
    const size_t lds = std::max(resize_down_lds_bytes(16, ncp, tile_w, h.stride), (size_t)4 * b.wave_floats * sizeof(float));
    if (lds > 64u * 1024u) return hipErrorInvalidValue;
    const dim3 grid(8u * b.xper + nt * b.gx, 1, batch);
    if (ages == 2) launch_resize_poly_a<2>(grid, lds, s, ratio, p, dw, dh, v, h, tile_w, ncp, b);
    else if (ages == 4) launch_resize_poly_a<4>(grid, lds, s, ratio, p, dw, dh, v, h, tile_w, ncp, b);
    else launch_resize_poly_a<6>(grid, lds, s, ratio, p, dw, dh, v, h, tile_w, ncp, b);
    return hipGetLastError();
}

// Integer-ratio down-sampling, two waves to a band's strip (round 4).
//
// resize_poly_kernel's geometry is what the memory system likes -- 256-column windows, one wave-wide row request per source row
// (profiles/tile_read_bench.hip: the bare access pattern of Gaussian 4096^2 -> 512^2 takes 15.6 us) -- but a band wave of its
// lives 27 us: a launch has about one wave per SIMD, and that wave's own instruction stream (the trips' packed arithmetic, then
// the horizontal pass) is the critical path.  Narrower strips give more and lighter waves and lose it all again on the wider
// halo (128-column windows: the bare pattern alone is 22.6 us).  So the strip stays and its work is cut in two along the
// columns: waves 2 p and 2 p + 1 of a workgroup take the left and right 128 columns of pair p's window as 8-byte lanes (half
// the arithmetic per trip each), both write their halves of each finished row into ONE ring in LDS, and after every fourth row
// the two share the horizontal pass -- one output pixel per lane, (row, column) = pair lane / strip width -- behind a single
// s_barrier (the ring holds eight rows, so nobody has to wait for the other's reads before writing on).  The barrier is a bare
// s_waitcnt lgkmcnt(0) + s_barrier: the rows requested for the next trips stay in flight across it.  Both pairs of a workgroup
// work on the same band (same number of trips and barriers).  Same taps in the same order: same roundings.
// Rows outside the regular range run as resize_down_kernel tiles in the launch's last workgroups, as before.
struct Poly2Bands {
    uint32_t ya, yb, rows, n_bands;
    uint32_t tw, n_strips;     // band path: output columns per strip (its source window is at most 256 columns), strips per row
    uint32_t n_wgx;            // workgroups per band (two strips each)
    uint32_t n_band_wgs;       // n_wgx * n_bands; the general tiles follow
    uint32_t gen_tw, gen_gx, gen_ncp;  // general tiles: resize_down_kernel's strip width, strips per row, padded window
    uint32_t n_gen;
    uint32_t ty0[6], th[6];
};
#define KC_POLY2_RING_PITCH 265u  // 256 columns + one pad per 32, odd
#ifndef KC_POLY2_NB
#define KC_POLY2_NB 1  // trips of rows in flight beyond the one in use: 1 / 2 / 3 measure the same or worse (profiles/r04_poly2_sweep.txt)
#endif
// (LDS writes of this wave done, then the workgroup's barrier; global loads stay in flight)
#define KC_POLY2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int A, int RT>
__global__ __launch_bounds__(256) void resize_poly2_kernel(const ResizePlanes P, uint32_t dw, uint32_t dh, TapsDev V, TapsDev H,
                                                           Poly2Bands B, uint32_t pair_floats, XcdOrder X)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float *__restrict__ src = P.src[blockIdx.z];
    float *__restrict__ dst = P.dst[blockIdx.z];
    const uint32_t spitch = P.spitch[blockIdx.z], dpitch = P.dpitch[blockIdx.z];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t wg = blockIdx.x;
    if (X.per) {
        // Workgroup id % 8 is the XCD.  XCD k takes the k-th eighth of the band workgroups in BAND-major order: whole bands, one
        // after the other -- the strips of a band (which share their halo columns) and the next band (which shares 40 of its 136
        // source rows at ratio 8) meet in one L2 while they are there, and every XCD streams one contiguous eighth of the plane.
        // (profiles/tile_read_bench.hip, the loads and the vertical arithmetic alone: 14.1 us like this, 22.2 in plain order;
        // KC_POLY2_XCD=2: the eighths in strip-major order, a strip pair with all its bands, as resize_poly_kernel has them)
        if (wg < 8u * X.per) {
            const uint32_t tile = (wg & 7u) * X.per + (wg >> 3);
            if (tile >= X.n) return;
            if (X.gy) {
                const uint32_t wx = __umulhi(tile, X.magic);
                wg = (tile - wx * X.gy) * B.n_wgx + wx;
            } else {
                wg = tile;
            }
        } else {
            wg = wg - 8u * X.per + B.n_band_wgs;
        }
    }
    if (wg >= B.n_band_wgs) {
        // rows near the border: the general form, all four waves on one tile
        const uint32_t g = wg - B.n_band_wgs;
        const uint32_t t = g / B.gen_gx, bx = g - t * B.gen_gx;
        const DownStrip S = resize_down_stage(lds, H, dw, B.gen_tw, 16u, B.gen_ncp, bx);
        resize_down_tile<4>(S, src, spitch, dst, dpitch, B.ty0[t], B.th[t], V, wave, lane);
        return;
    }
    // ---- a band workgroup: strips 2 wgx, 2 wgx + 1 of band `band` (an odd strip count: the last strip twice, same values) ----
    const uint32_t band = wg / B.n_wgx, wgx = wg - band * B.n_wgx;
    const uint32_t pair = wave >> 1, half = wave & 1u;
    const uint32_t strip = min(2u * wgx + pair, B.n_strips - 1u);
    const uint32_t x0 = strip * B.tw, x1 = min(x0 + B.tw, dw), tw = x1 - x0;
    const uint32_t c0 = H.left[x0] & ~3u;
    const uint32_t ncols = H.left[x1 - 1] + H.count[x1 - 1] - c0;  // <= 256 (host-checked)
    const uint32_t npairs = (ncols + 1u) / 2u;
    float *ring = lds + pair * pair_floats;                        // 8 rows x KC_POLY2_RING_PITCH
    uint32_t *hl = reinterpret_cast<uint32_t *>(ring + 8u * KC_POLY2_RING_PITCH);
    uint32_t *hn = hl + 128;
    float *hw = reinterpret_cast<float *>(hn + 128);               // tw x hsp
    const uint32_t hsp = H.stride | 1u;  // odd pitch: the lanes' weight rows start on different banks
    const uint32_t yf = B.ya + B.rows * band;
    const uint32_t ROWS = min(B.rows, B.yb - yf);
    const uint32_t sp2 = spitch / 2u;
    const uint32_t pl = half * 64u + lane;  // lane of the pair
    const bool p_ok = pl < npairs;
    const uint32_t pq = min(pl, npairs - 1u);
    // (a column pair's second column may be the first one past the window: inside the row or its padding, never past the pitch)
    const f2 *col = reinterpret_cast<const f2 *>(src + c0 + (size_t)V.left[yf] * spitch) + pq;
    // KC_POLY2_NB trips of rows in flight or in use: buffer b holds trips b, b + NB, ... -- named statically, so that the wait for
    // a trip's rows leaves the younger trips' loads in flight
    constexpr int NB = KC_POLY2_NB;
    const uint32_t TRIPS = ROWS + A - 1;
    f2 pb[NB][RT];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int u = 0; u < RT; ++u) pb[b][u] = col[(size_t)(min((uint32_t)b, TRIPS - 1u) * RT + u) * sp2];
    // the pair's copy of its strip's horizontal taps (in flight together with the first rows)
    for (uint32_t i = pl; i < tw; i += 128u) {
        hl[i] = H.left[x0 + i] - c0;
        hn[i] = H.count[x0 + i];
    }
    for (uint32_t i = pl; i < tw * H.stride; i += 128u) {
        const uint32_t x = i / H.stride, j = i - x * H.stride;
        hw[x * hsp + j] = H.w[(size_t)x0 * H.stride + i];
    }
    // the A RT weights, two to a vector register pair, broadcast by op_sel inside the loop (as in resize_poly_kernel)
    f2 Wp[A][RT / 2];
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int u = 0; u < RT; u += 2)
            Wp[a][u / 2] = f2{ V.w[(size_t)B.ya * V.stride + a * RT + u], V.w[(size_t)B.ya * V.stride + a * RT + u + 1] };
    auto mad = [&](f2 &sum, const f2 &p, int a, int u) {
        asm volatile("" : "+v"(Wp[a][u / 2]));
        const f2 wp = Wp[a][u / 2];
        sum += p * ((u & 1) ? __builtin_shufflevector(wp, wp, 1, 1) : __builtin_shufflevector(wp, wp, 0, 0));
    };
    const uint32_t rj = 2u * pq + ((2u * pq) >> 5);  // where this lane's pair goes in a ring row (a pair never straddles a pad)
    // which pixel of four finished rows this lane takes in the horizontal pass: all four rows at once for strips of up to 32
    // columns, two for up to 64, one row of up to 128 columns at a time beyond
    const uint32_t rows_per_pass = tw <= 32u ? 4u : tw <= 64u ? 2u : 1u;  // (uniform)
    const uint32_t hx = rows_per_pass == 4u ? (pl & 31u) : rows_per_pass == 2u ? (pl & 63u) : pl;
    const uint32_t hr = rows_per_pass == 4u ? (pl >> 5) : rows_per_pass == 2u ? (pl >> 6) : 0u;
    const bool h_ok = hx < tw;
    const uint32_t hxs = h_ok ? hx : 0u;
    f2 acc[A];
#pragma unroll
    for (int a = 0; a < A; ++a) acc[a] = f2{ 0.0f, 0.0f };
    KC_POLY2_BARRIER();  // the taps are staged
    const uint32_t hcount = hn[hxs], h0 = hl[hxs];
    const float *hwt = hw + hxs * hsp;
    const uint32_t nu = (uint32_t)__builtin_amdgcn_readfirstlane((int)hcount);
    const bool uniform = (nu & 3u) == 0u && __builtin_amdgcn_ballot_w64(hcount != nu) == 0ull;
    for (uint32_t cb = 0; cb < TRIPS; cb += NB) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint32_t c = cb + b;  // (up to NB - 1 trips past the last one do nothing: leaving the loop from its middle would join
                                    // paths with different numbers of loads in flight, and the waits would be for all of them)
        f2 (&p)[RT] = pb[b];
        // age a holds row c - a of the band
        if (c >= (uint32_t)(A - 1) && c < ROWS) {
#pragma unroll
            for (int u = 0; u < RT; ++u)
#pragma unroll
                for (int a = 0; a < A; ++a) mad(acc[a], p[u], a, u);
        } else {
#pragma unroll
            for (int a = 0; a < A; ++a)
                if (c >= (uint32_t)a && c - (uint32_t)a < ROWS) {
#pragma unroll
                    for (int u = 0; u < RT; ++u) mad(acc[a], p[u], a, u);
                }
        }
        // the buffer's next trip (past the last one: that one again, so that the number of loads in flight is the same on every path)
        {
            const uint32_t cn = min(c + (uint32_t)NB, TRIPS - 1u);
#pragma unroll
            for (int u = 0; u < RT; ++u) {
                asm volatile("" : "+v"(p[u]));  // (after this trip's last use of the row, not in a register of its own)
                p[u] = col[(size_t)(cn * RT + u) * sp2];
            }
        }
        if (c >= (uint32_t)(A - 1) && c < TRIPS) {
            const uint32_t k = c - (uint32_t)(A - 1);  // this row of the band is complete
            if (p_ok) {
                float *o = ring + (k & 7u) * KC_POLY2_RING_PITCH + rj;
                o[0] = acc[A - 1].x;
                o[1] = acc[A - 1].y;
            }
            if ((k & 3u) == 3u) {
                KC_POLY2_BARRIER();  // both halves of the four rows are in the ring (and everybody is done with the four before)
                const float *rows4 = ring + (k & 4u) * KC_POLY2_RING_PITCH;
                for (uint32_t r0 = 0; r0 < 4u; r0 += rows_per_pass) {
                    const float *row = rows4 + (r0 + hr) * KC_POLY2_RING_PITCH;
                    float t = 0.0f;
                    if (uniform) {
                        for (uint32_t j0 = 0; j0 < nu; j0 += 4u) {
                            float pv[4], wt[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const uint32_t idx = h0 + j0 + u;
                                pv[u] = row[idx + (idx >> 5)];
                                wt[u] = hwt[j0 + u];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) t += pv[u] * wt[u];
                        }
                    } else {
                        for (uint32_t j0 = 0; j0 < hcount; j0 += 4u) {
                            float pv[4], wt[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const uint32_t jj = min(j0 + u, hcount - 1u);
                                const uint32_t idx = h0 + jj;
                                pv[u] = row[idx + (idx >> 5)];
                                wt[u] = hwt[jj];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) t += j0 + u < hcount ? pv[u] * wt[u] : -0.0f;  // -0.0 leaves the sum as it is
                        }
                    }
                    if (h_ok) dst[(size_t)(yf + k - 3u + r0 + hr) * dpitch + x0 + hx] = clamp01_nan_passthrough(t);
                }
            }
        }
#pragma unroll
        for (int a = A - 1; a > 0; --a) acc[a] = acc[a - 1];
        acc[0] = f2{ 0.0f, 0.0f };
      }
    }
}

template <int A>
static void launch_resize_poly2_a(dim3 grid, size_t lds, hipStream_t s, uint32_t rt, const ResizePlanes &p, uint32_t dw, uint32_t dh,
                                  TapsDev v, TapsDev h, const Poly2Bands &b, uint32_t pair_floats, const XcdOrder &x)
{
    if (rt == 2) resize_poly2_kernel<A, 2><<<grid, 256, lds, s>>>(p, dw, dh, v, h, b, pair_floats, x);
    else if (rt == 4) resize_poly2_kernel<A, 4><<<grid, 256, lds, s>>>(p, dw, dh, v, h, b, pair_floats, x);
    else resize_poly2_kernel<A, 8><<<grid, 256, lds, s>>>(p, dw, dh, v, h, b, pair_floats, x);
}

// tw: output columns per band strip (host-checked: every strip's source window, from its first column rounded down to a multiple
// of 4, is at most 256 columns); gen_tw / gen_ncp: resize_down_kernel's tile for the border rows.
hipError_t launch_resize_poly2(const ResizePlanes &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h, uint32_t tw,
                               uint32_t gen_tw, uint32_t gen_ncp, uint32_t reg_a, uint32_t reg_b, uint32_t ages, uint32_t ratio, bool xcd,
                               hipStream_t s)
{
    if (dw == 0 || dh == 0) return hipSuccess;
    if (batch < 1 || batch > 4) return hipErrorInvalidValue;
    if (tw == 0 || tw > 128 || gen_tw == 0 || gen_tw > 64 || gen_ncp % 4 != 0 || gen_ncp > 256 || reg_a > reg_b || reg_b > dh) return hipErrorInvalidValue;
    if ((ages != 2 && ages != 4 && ages != 6) || (ratio != 2 && ratio != 4 && ratio != 8)) return hipErrorInvalidValue;
    Poly2Bands b{};
    b.ya = reg_a;
    b.yb = reg_a + (reg_b - reg_a) / 4u * 4u;
    if (b.yb == b.ya) return hipErrorInvalidValue;
    b.tw = tw;
    b.n_strips = (dw + tw - 1) / tw;
    b.n_wgx = (b.n_strips + 1u) / 2u;
    // band height as resize_poly_kernel chooses it (a launch lasts as long as one wave lives), for twice the waves per strip
    static const uint32_t rows_env = std::getenv("KC_POLY_ROWS") ? std::max(4u, (uint32_t)std::atoi(std::getenv("KC_POLY_ROWS")) / 4u * 4u) : 0u;
    b.rows = 12u;
    {
        // (2048^2 -> 256^2 Gaussian: 16.5 / 14.7 us with 4 / 8 rows = 1280 / 640 waves)
        const uint64_t regular = b.yb - b.ya, waves4 = 4u * b.n_wgx * ((regular + 3u) / 4u) * (uint64_t)batch,
                       waves8 = 4u * b.n_wgx * ((regular + 7u) / 8u) * (uint64_t)batch;
        if (waves4 <= 1100u) b.rows = 4u;
        else if (waves8 <= 2200u) b.rows = 8u;
        // (and where 12-row bands give six waves per SIMD or more -- RGBA launches from 4096^2 on -- taller bands re-read less:
        // 4096^2 -> 512^2 RGBA 68.0 -> 66.0 us, 8192^2 -> 1024^2 RGBA 286.6 -> 271.1; 2048^2 -> 256^2 RGBA 24.9 -> 38.2, one plane
        // at 4096^2 25.3 -> 30.3: profiles/r04_poly2_rows_rgba.txt, r04_poly2_sweep.txt)
        else if (4u * b.n_wgx * ((regular + 11u) / 12u) * (uint64_t)batch >= 6000u) b.rows = 24u;
    }
    if (rows_env) b.rows = rows_env;
    b.n_bands = (b.yb - b.ya + b.rows - 1) / b.rows;
    b.n_band_wgs = b.n_wgx * b.n_bands;
    b.gen_tw = gen_tw;
    b.gen_ncp = gen_ncp;
    b.gen_gx = (dw + gen_tw - 1) / gen_tw;
    uint32_t nt = 0;
    auto add_tiles = [&](uint32_t y0, uint32_t y1) {
        for (uint32_t y = y0; y < y1; y += 16u) {
            if (nt == 6) return false;
            b.ty0[nt] = y;
            b.th[nt] = std::min(16u, y1 - y);
            ++nt;
        }
        return true;
    };
    if (!add_tiles(0, b.ya) || !add_tiles(b.yb, dh)) return hipErrorInvalidValue;
    b.n_gen = nt;
    const uint32_t pair_floats = (8u * KC_POLY2_RING_PITCH + 256u + tw * (h.stride | 1u) + 3u) / 4u * 4u;
    const size_t lds = std::max((size_t)2 * pair_floats * sizeof(float), resize_down_lds_bytes(16, gen_ncp, gen_tw, h.stride));
    if (lds > 64u * 1024u) return hipErrorInvalidValue;
    static const int xcd_env = std::getenv("KC_POLY2_XCD") ? std::atoi(std::getenv("KC_POLY2_XCD")) : -1;  // 0: plain order, 1: band-major eighths, 2: strip-major
    (void)xcd;
    XcdOrder x{ 0, 0, 0, 0 };
    if (xcd_env == 2) {
        x = xcd_order(b.n_wgx, b.n_bands, true);
    } else if (xcd_env != 0 && b.n_band_wgs >= 16u) {
        x.per = (b.n_band_wgs + 7u) / 8u;
        x.n = b.n_band_wgs;
    }
    dim3 grid((x.per ? 8u * x.per : b.n_band_wgs) + nt * b.gen_gx, 1, batch);
    if (ages == 2) launch_resize_poly2_a<2>(grid, lds, s, ratio, p, dw, dh, v, h, b, pair_floats, x);
    else if (ages == 4) launch_resize_poly2_a<4>(grid, lds, s, ratio, p, dw, dh, v, h, b, pair_floats, x);
    else launch_resize_poly2_a<6>(grid, lds, s, ratio, p, dw, dh, v, h, b, pair_floats, x);
    return hipGetLastError();
}

template <int K>
static hipError_t launch_resize_chain_k(const ChainProgram &p, dim3 grid, size_t lds, hipStream_t s, uint32_t dw,
                                        uint32_t dh, TapsDev v, TapsDev h, uint32_t tile_w, uint32_t tile_h, uint32_t ncp)
{
    switch (h.stride) {
    case 1: resize_chain_kernel<K, 1><<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, tile_h, ncp); break;
    case 2: resize_chain_kernel<K, 2><<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, tile_h, ncp); break;
    case 3: resize_chain_kernel<K, 3><<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, tile_h, ncp); break;
    case 4: resize_chain_kernel<K, 4><<<grid, 256, lds, s>>>(p, dw, dh, v, h, tile_w, tile_h, ncp); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_resize_chain(const ChainProgram &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h,
                               uint32_t tile_w, uint32_t tile_h, uint32_t ncp, hipStream_t s)
{
    if (dw == 0 || dh == 0) return hipSuccess;
    if (batch < 1 || batch > KC_CHAIN_MAX_BATCH || p.n_ops < 1 || p.n_ops > KC_CHAIN_MAX_OPS) return hipErrorInvalidValue;
    if (tile_w % 4 != 0 || tile_w > 1024 || 256u % (tile_w / 4) != 0 || tile_h > 64) return hipErrorInvalidValue;
    const size_t lds = resize_lds_bytes(tile_h, ncp, v.stride, tile_w, h.stride);
    dim3 grid((dw + tile_w - 1) / tile_w, (dh + tile_h - 1) / tile_h, batch);
    switch (p.n_in) {
    case 1: return launch_resize_chain_k<1>(p, grid, lds, s, dw, dh, v, h, tile_w, tile_h, ncp);
    case 2: return launch_resize_chain_k<2>(p, grid, lds, s, dw, dh, v, h, tile_w, tile_h, ncp);
    case 3: return launch_resize_chain_k<3>(p, grid, lds, s, dw, dh, v, h, tile_w, tile_h, ncp);
    case 4: return launch_resize_chain_k<4>(p, grid, lds, s, dw, dh, v, h, tile_w, tile_h, ncp);
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------
// Integer-ratio up-sampling (upsample.h, upsample_chain.inc): the fused resample + chain kernel and the plain resize
// kernel for out = R x in.  Replaces resize_chain_kernel / resize_lds_kernel where the host's check holds; same
// operations per sample, bit-identical output.
// ------------------------------------------------------------------------------------------
#include "upsample_chain.inc"

// Rows per thread (one trip per workgroup, see upsample_chain.inc).  KC_UP_RU at build time overrides (tuning).
#ifndef KC_UP_RU
#define KC_UP_RU 4
#endif
static_assert(KC_UP_RU == KC_UPSAMPLE_ROWS, "kc_internal.hpp sizes the tiles for this many rows per thread");

template <int K, int RU>
struct UpInterpreted {  // the chain as the step interpreter runs it (first sightings, no hiprtc)
    const ChainProgram &P;
    __device__ __forceinline__ void operator()(const f4 (&in)[K][RU], f4 (&acc)[RU]) const { chain_run<K, RU, 0>(P, blockIdx.z, in, acc); }
};

template <int RU>
struct UpStore {  // no chain: the resampled plane itself is the result
    __device__ __forceinline__ void operator()(const f4 (&in)[1][RU], f4 (&acc)[RU]) const
    {
#pragma unroll
        for (int u = 0; u < RU; ++u) acc[u] = in[0][u];
    }
};

#ifdef KC_UP_HARDWIRE  // tuning builds only: config #2's program ((A + U) * A - U) in place of the interpreter
template <int K, int RU>
struct UpHardwired {
    __device__ __forceinline__ void operator()(const f4 (&in)[K][RU], f4 (&acc)[RU]) const
    {
#pragma unroll
        for (int u = 0; u < RU; ++u) acc[u] = (in[0][u] + in[K - 1][u]) * in[0][u] - in[K - 1][u];
    }
};
#endif

#ifdef KC_UP_SGPR
#define KC_UP_ATTR __attribute__((amdgpu_num_sgpr(KC_UP_SGPR)))
#else
#define KC_UP_ATTR
#endif
template <int K, int T, bool WIDE, bool HALF>  // HALF: the horizontal ratio is 2 (upsample.h)
__global__ __launch_bounds__(256) KC_UP_ATTR void upsample_chain_kernel(const ChainProgram P, const UpsampleArgs U)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef KC_UP_HARDWIRE
    upsample_chain_tile<K, T, KC_UP_RU, WIDE, 0u, HALF>(P, U, lds, UpHardwired<K, KC_UP_RU>{});
#else
    upsample_chain_tile<K, T, KC_UP_RU, WIDE, 0u, HALF>(P, U, lds, UpInterpreted<K, KC_UP_RU>{ P });
#endif
}

template <int T, bool WIDE, bool NTS, bool HALF>  // NTS: the resampled planes are stored nontemporal (cache_policy_mask)
__global__ __launch_bounds__(256) void upsample_kernel(const UpsamplePlanes P, const UpsampleArgs U)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    upsample_chain_tile<1, T, KC_UP_RU, WIDE, (NTS ? 0x100u : 0u), HALF>(P, U, lds, UpStore<KC_UP_RU>{});
}

template <int K, bool WIDE>
static hipError_t launch_upsample_chain_k(const ChainProgram &p, const UpsampleArgs &u, dim3 grid, size_t lds, hipStream_t s)
{
    const bool half = u.H.ratio == 2;
    switch (u.H.taps) {
    case 1:
        if (half) upsample_chain_kernel<K, 1, WIDE, true><<<grid, 256, lds, s>>>(p, u);
        else upsample_chain_kernel<K, 1, WIDE, false><<<grid, 256, lds, s>>>(p, u);
        break;
    case 3:
        if (half) upsample_chain_kernel<K, 3, WIDE, true><<<grid, 256, lds, s>>>(p, u);
        else upsample_chain_kernel<K, 3, WIDE, false><<<grid, 256, lds, s>>>(p, u);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_upsample_chain(const ChainProgram &p, int batch, const UpsampleArgs &u, hipStream_t s)
{
    if (u.H.n_out == 0 || u.V.n_out == 0) return hipSuccess;
    if (!upsample_args_ok(u, batch) || p.n_ops < 1 || p.n_ops > KC_CHAIN_MAX_OPS) return hipErrorInvalidValue;
    const size_t lds = upsample_lds_bytes(u);
    const dim3 grid = upsample_grid(u, batch);
    const bool wide = u.tile_w == 1024;
    switch (p.n_in) {
    case 1: return wide ? launch_upsample_chain_k<1, true>(p, u, grid, lds, s) : launch_upsample_chain_k<1, false>(p, u, grid, lds, s);
    case 2: return wide ? launch_upsample_chain_k<2, true>(p, u, grid, lds, s) : launch_upsample_chain_k<2, false>(p, u, grid, lds, s);
    case 3: return wide ? launch_upsample_chain_k<3, true>(p, u, grid, lds, s) : launch_upsample_chain_k<3, false>(p, u, grid, lds, s);
    case 4: return wide ? launch_upsample_chain_k<4, true>(p, u, grid, lds, s) : launch_upsample_chain_k<4, false>(p, u, grid, lds, s);
    default: return hipErrorInvalidValue;
    }
}

template <bool WIDE, bool NTS>
static hipError_t launch_upsample_w(const UpsamplePlanes &p, const UpsampleArgs &u, dim3 grid, size_t lds, hipStream_t s)
{
#define KC_UP_LAUNCH(T)                                                                 \
    do {                                                                               \
        if (u.H.ratio == 2) upsample_kernel<T, WIDE, NTS, true><<<grid, 256, lds, s>>>(p, u);  \
        else upsample_kernel<T, WIDE, NTS, false><<<grid, 256, lds, s>>>(p, u);        \
    } while (0)
    switch (u.H.taps) {
    case 1: KC_UP_LAUNCH(1); break;
    case 3: KC_UP_LAUNCH(3); break;
    case 5: KC_UP_LAUNCH(5); break;
    case 7: KC_UP_LAUNCH(7); break;
    default: return hipErrorInvalidValue;
    }
#undef KC_UP_LAUNCH
    return hipGetLastError();
}

hipError_t launch_upsample(const UpsamplePlanes &p, int batch, const UpsampleArgs &u, hipStream_t s)
{
    if (u.H.n_out == 0 || u.V.n_out == 0) return hipSuccess;
    if (!upsample_args_ok(u, batch)) return hipErrorInvalidValue;
    const size_t lds = upsample_lds_bytes(u);
    const dim3 grid = upsample_grid(u, batch);
    const bool nts = (p.nt_mask & 0x100u) != 0;
    if (u.tile_w == 1024) return nts ? launch_upsample_w<true, true>(p, u, grid, lds, s) : launch_upsample_w<true, false>(p, u, grid, lds, s);
    return nts ? launch_upsample_w<false, true>(p, u, grid, lds, s) : launch_upsample_w<false, false>(p, u, grid, lds, s);
}

// ------------------------------------------------------------------------------------------
// HeightToNormal: src/node/height_to_normal.rs:16-77 with the toroidal wrap of
// src/node/process_shared.rs:31-65; nalgebra 0.29 normalize = v / sqrt((x*x + y*y) + z*z).
// 4 B read + 12 B written per pixel (alpha is a constant plane).
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ void vnorm3(float x, float y, float z, float &ox, float &oy, float &oz)
{
    const float n = sqrtf((x * x + y * y) + z * z);
    ox = x / n;
    oy = y / n;
    oz = z / n;
}

// 0.0f / n for a norm n that is never zero here (n >= 1/width > 0): +0 unless n is NaN.  Saves two
// of the nine IEEE divisions per pixel; this kernel is bound by divide / sqrt issue, not by HBM.
static __device__ __forceinline__ float zero_over(float n) { return n != n ? n : 0.0f; }

// Several IEEE divisions by one denominator.  This is the compiler's own correctly rounded f32
// division (v_div_scale / v_rcp / Newton steps / v_div_fmas / v_div_fixup) with the steps that depend
// only on the denominator done once -- valid where v_div_scale would not rescale and v_div_fixup
// would not intervene: b normal with a normal reciprocal, a == 0 or |a| >= 2^-103, a / b normal and
// exponent(a) - exponent(b) < 96.  h2n_px establishes those bounds before taking this path.
struct SharedDenominator {
    float nb, r;  // -b, reciprocal after one Newton step
};

static __device__ __forceinline__ SharedDenominator shared_denominator(float b)
{
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    return { -b, __builtin_fmaf(e, r0, r0) };
}

template <bool MAY_BE_ZERO = true>
static __device__ __forceinline__ float divide_by(const SharedDenominator &d, float a)
{
    const float m = a * d.r;
    const float f2 = __builtin_fmaf(d.nb, m, a);
    const float f3 = __builtin_fmaf(f2, d.r, m);
    const float f4 = __builtin_fmaf(d.nb, f3, a);
    const float q = __builtin_fmaf(f4, d.r, f3);
    // b > 0: the quotient has a's sign; for a == -0 the steps above give +0, so put the sign back
    return MAY_BE_ZERO ? __builtin_copysignf(q, a) : q;
}

// sqrt for normal x without the compiler's denormal scaling and +-1 ulp fix-up: the rsq / Newton /
// residual sequence LLVM itself uses when denormals are flushed.  Correctly rounded on [2^-96, 2^100)
// (every value checked against sqrtf: profiles/exact_math_check.hip).
static __device__ __forceinline__ float sqrt_normal(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    const float h0 = y * 0.5f;
    const float e = __builtin_fmaf(-h0, s0, 0.5f);
    const float h = __builtin_fmaf(h0, e, h0);
    const float s = __builtin_fmaf(s0, e, s0);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}

static __device__ __forceinline__ void h2n_px(float px, float up, float left, float pdx, float pdy, float &r, float &g,
                                              float &b)
{
    // tangent = (pdx, 0, px - left) / |.|, bitangent = (0, pdy, up - px) / |.|;
    // |v| = sqrt((x*x + y*y) + z*z) and x*x + 0*0 == x*x exactly
    const float tz0 = px - left, bz0 = up - px;
    const float q1 = pdx * pdx + tz0 * tz0, q2 = pdy * pdy + bz0 * bz0;
    // Height steps that are 0 or within [2^-40, 2^7] (and 2^-16 <= pdx, pdy <= 1: sizes are at most
    // 65535) keep every operation below inside the bounds of sqrt_normal / SharedDenominator:
    // q1, q2 in [2^-32, 2^15], n1, n2 in [2^-16, 2^7.5], tangent parts in {0} u [2^-47.5, 1], cross
    // products in {0} u [2^-71, 1], cz >= 2^-47, so |cross|^2 in [2^-94, 3].
    const float atz = fabsf(tz0), abz = fabsf(bz0);
    const bool tame = (tz0 == 0.0f || (atz >= 0x1p-40f && atz <= 0x1p7f)) && (bz0 == 0.0f || (abz >= 0x1p-40f && abz <= 0x1p7f));
    float nx, ny, nz;
    if (tame) {
        const SharedDenominator d1 = shared_denominator(sqrt_normal(q1)), d2 = shared_denominator(sqrt_normal(q2));
        const float tx = divide_by<false>(d1, pdx), tz = divide_by(d1, tz0);
        const float by = divide_by<false>(d2, pdy), bz = divide_by(d2, bz0);
        const float ty = 0.0f, bx = 0.0f;  // 0 / n
        const float cx = ty * bz - tz * by;
        const float cy = tz * bx - tx * bz;
        const float cz = tx * by - ty * bx;
        const SharedDenominator d3 = shared_denominator(sqrt_normal((cx * cx + cy * cy) + cz * cz));
        nx = divide_by(d3, cx);
        ny = divide_by(d3, cy);
        nz = divide_by<false>(d3, cz);  // cz = tx * by > 0
    } else {
        const float n1 = sqrtf(q1), n2 = sqrtf(q2);
        const float tx = pdx / n1, ty = zero_over(n1), tz = tz0 / n1;
        const float bx = zero_over(n2), by = pdy / n2, bz = bz0 / n2;
        const float cx = ty * bz - tz * by;
        const float cy = tz * bx - tx * bz;
        const float cz = tx * by - ty * bx;
        vnorm3(cx, cy, cz, nx, ny, nz);
    }
    r = nx * 0.5f + 0.5f;
    g = ny * 0.5f + 0.5f;
    b = nz * 0.5f + 0.5f;
}

// The same arithmetic for the 4 pixels of a quad at once, written on 4-wide vectors so that the Newton / residual
// steps of the shared-denominator division and of the square root become packed instructions (v_pk_fma_f32,
// v_pk_mul_f32: two pixels per instruction) -- this kernel is bound by vector-instruction issue, not by HBM.
// Element by element these are exactly the operations of h2n_px's `tame` branch (same instructions, same order);
// a quad with any pixel outside that range goes through h2n_px pixel by pixel.
typedef float f2 __attribute__((ext_vector_type(2)));
template <class V> static __device__ __forceinline__ V fmaV(V a, V b, V c) { return __builtin_elementwise_fma(a, b, c); }
template <class V> static __device__ __forceinline__ V splatV(float v)
{
    V o;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(V) / sizeof(float)); ++i) o[i] = v;
    return o;
}
template <class V> static __device__ __forceinline__ V rsqV(V x)
{
    V o;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(V) / sizeof(float)); ++i) o[i] = __builtin_amdgcn_rsqf(x[i]);
    return o;
}
template <class V> static __device__ __forceinline__ V copysignV(V mag, V sgn)
{
    V o;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(V) / sizeof(float)); ++i) o[i] = __builtin_copysignf(mag[i], sgn[i]);
    return o;
}
template <class V> struct SharedDenominatorV {
    V nb, r;
};
// The denominator-only part of the division by n = sqrt_normal(x), with the reciprocal seeded by the rsq the square root
// starts from anyway instead of a separate v_rcp_f32 of n: y = rsq(x) is 1 / n to ~2^-22, one Newton step on n takes it to
// the same ~2^-45 the rcp-seeded step reaches, and the quotient's correction steps are the same.  Transcendental
// instructions run at a quarter of the packed-math rate: this halves them (6 -> 3 per pixel).  Checked against a / sqrtf(x)
// over 3 x 2^34 (a, x) pairs in the ranges h2n_quad establishes (profiles/exact_math_check.hip, r02_exact_math_check.txt).
template <class V> static __device__ __forceinline__ SharedDenominatorV<V> sqrt_denominatorV(V x)
{
    const V half = splatV<V>(0.5f), one = splatV<V>(1.0f);
    const V y = rsqV(x);
    const V s0 = x * y;
    const V h0 = y * half;
    const V e = fmaV(-h0, s0, half);
    const V h = fmaV(h0, e, h0);
    const V s = fmaV(s0, e, s0);
    const V d = fmaV(-s, s, x);
    const V n = fmaV(d, h, s);  // sqrt_normal(x)
    const V er = fmaV(-n, y, one);
    return { -n, fmaV(er, y, y) };
}

template <bool MAY_BE_ZERO, class V>
static __device__ __forceinline__ V divide_byV(const SharedDenominatorV<V> &d, V a)
{
    const V m = a * d.r;
    const V f2_ = fmaV(d.nb, m, a);
    const V f3 = fmaV(f2_, d.r, m);
    const V f4_ = fmaV(d.nb, f3, a);
    const V q = fmaV(f4_, d.r, f3);
    return MAY_BE_ZERO ? copysignV(q, a) : q;
}

// The tame path on V = 2 or 4 pixels: element by element exactly the operations of h2n_px's `tame` branch.
template <class V>
static __device__ __forceinline__ void h2n_fast(V tz0, V bz0, float pdx, float pdy, V &r, V &g, V &b)
{
    const V vdx = splatV<V>(pdx), vdy = splatV<V>(pdy), half = splatV<V>(0.5f);
    const V q1 = vdx * vdx + tz0 * tz0, q2 = vdy * vdy + bz0 * bz0;
    const SharedDenominatorV<V> d1 = sqrt_denominatorV(q1), d2 = sqrt_denominatorV(q2);
    const V tx = divide_byV<false>(d1, vdx), tz = divide_byV<true>(d1, tz0);
    const V by = divide_byV<false>(d2, vdy), bz = divide_byV<true>(d2, bz0);
    // t = (tx, 0, tz), b = (0, by, bz): the cross product's products with the two zero components (0 / n = +0) vanish.
    //   cx = 0 * bz - tz * by = -(tz * by),  cy = tz * 0 - tx * bz = -(tx * bz),  cz = tx * by - 0 * 0 = tx * by
    // exactly, for the finite values of this path -- except the SIGN of a zero result (+-0 - +-0), which cannot reach the
    // output: cx and cy enter as squares and as (+-0 / n) * 0.5 + 0.5 = 0.5.
    const V cx = -(tz * by);
    const V cy = -(tx * bz);
    const V cz = tx * by;
    const SharedDenominatorV<V> d3 = sqrt_denominatorV((cx * cx + cy * cy) + cz * cz);
    const V nx = divide_byV<true>(d3, cx), ny = divide_byV<true>(d3, cy), nz = divide_byV<false>(d3, cz);
    r = nx * half + half;
    g = ny * half + half;
    b = nz * half + half;
}

static __device__ __forceinline__ bool tame1(float d)
{
    const float a = fabsf(d);
    return d == 0.0f || (a >= 0x1p-40f && a <= 0x1p7f);
}

// px, up, left: the quad's heights, the heights above them, the heights to their left
static __device__ __forceinline__ void h2n_quad(f4 px, f4 up, f4 left, float pdx, float pdy, f4 &r, f4 &g, f4 &b)
{
    const f4 tz0 = px - left, bz0 = up - px;
    const bool tame = tame1(tz0.x) && tame1(tz0.y) && tame1(tz0.z) && tame1(tz0.w) && tame1(bz0.x) && tame1(bz0.y) &&
                      tame1(bz0.z) && tame1(bz0.w);
    if (!tame) {
        float rr[4], gg[4], bb[4];
        h2n_px(px.x, up.x, left.x, pdx, pdy, rr[0], gg[0], bb[0]);
        h2n_px(px.y, up.y, left.y, pdx, pdy, rr[1], gg[1], bb[1]);
        h2n_px(px.z, up.z, left.z, pdx, pdy, rr[2], gg[2], bb[2]);
        h2n_px(px.w, up.w, left.w, pdx, pdy, rr[3], gg[3], bb[3]);
        r = f4{ rr[0], rr[1], rr[2], rr[3] };
        g = f4{ gg[0], gg[1], gg[2], gg[3] };
        b = f4{ bb[0], bb[1], bb[2], bb[3] };
        return;
    }
    // (as two pairs in sequence the fast path fits 62 VGPRs = 8 waves per SIMD, and is no faster: 50.1-50.5 against 51.3 us,
    // profiles/r03_h2n_ab.txt -- the kernel is bound by vector issue and its hazard nops, not by occupancy)
    h2n_fast<f4>(tz0, bz0, pdx, pdy, r, g, b);
}

// BAND = false: the whole plane, rows wrap around (row -1 = row h - 1).  BAND = true: a row band -- `hgt` holds
// h + 1 rows, the band's rows preceded by the row above its first one (the caller's halo: the previous band's last
// row, or the image's last row for the band that starts at row 0); `full_h` is the height of the whole image, which
// is what the bitangent's 1 / height means (src/node/height_to_normal.rs:38).
// TILED: a workgroup is 2^tq column quads x 256 / 2^tq rows instead of 256 consecutive quads of one row, and the grid is
// (column block, row group) with the column block fastest.  Workgroups go to the 8 XCDs in turn (id % 8), so with a multiple of
// 8 column blocks per row a column block stays on ONE XCD all the way down the image: the row above, which every pixel reads,
// is in this workgroup or was loaded a moment ago by the same XCD -- an L2 hit instead of a second trip over the fabric for
// the whole plane (4096^2: 51.0 -> 43.7 us, 0.66 -> 0.77 of the HBM peak; profiles/r03_h2n_tiled_ab.txt).
template <bool BAND, bool NT, bool TILED>  // NT: the three result planes do not fit the Infinity Cache (cache_policy_mask)
__global__ __launch_bounds__(256) void height_to_normal_kernel(const float *__restrict__ hgt, uint32_t hpitch,
                                                               uint32_t w, uint32_t h, uint32_t full_h,
                                                               float *__restrict__ nx, float *__restrict__ ny,
                                                               float *__restrict__ nz, uint32_t opitch, uint32_t tq)
{
    const uint32_t row_units = (w + 3) / 4;
    const uint32_t total = row_units * h;
    const float pdx = 1.0f / (float)w;
    const float pdy = 1.0f / (float)full_h;
    auto pixel_quad = [&](uint32_t y, uint32_t q) {
        const uint32_t yc = BAND ? y + 1 : y;                          // row of this pixel in `hgt`
        const uint32_t yu = BAND ? y : (y == 0 ? h - 1 : y - 1);       // row above it
        const float *rowp = hgt + (size_t)yc * hpitch;
        const f4 cur = *reinterpret_cast<const f4 *>(rowp + 4 * q);
        const f4 upv = *reinterpret_cast<const f4 *>(hgt + (size_t)yu * hpitch + 4 * q);
        const float lft = q == 0 ? rowp[w - 1] : rowp[4 * q - 1];
        f4 r, g, b;
        h2n_quad(cur, upv, f4{ lft, cur.x, cur.y, cur.z }, pdx, pdy, r, g, b);
        const size_t o = (size_t)y * opitch + 4 * q;
        st_policy<NT>(reinterpret_cast<f4 *>(nx + o), r);
        st_policy<NT>(reinterpret_cast<f4 *>(ny + o), g);
        st_policy<NT>(reinterpret_cast<f4 *>(nz + o), b);
    };
    if (TILED) {  // the grid covers the image: one quad per thread
        const uint32_t q = (blockIdx.x << tq) + (threadIdx.x & ((1u << tq) - 1u)), y = (blockIdx.y << (8u - tq)) + (threadIdx.x >> tq);
        if (q < row_units && y < h) pixel_quad(y, q);
    } else {
        for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
            const uint32_t y = idx / row_units;
            pixel_quad(y, idx - y * row_units);
        }
    }
}

// h = rows to produce; band != 0: `hgt` has h + 1 rows (halo row first) and full_h is the whole image's height.
hipError_t launch_height_to_normal(const float *hgt, uint32_t hpitch, uint32_t w, uint32_t h, uint32_t full_h, int band,
                                   float *nx, float *ny, float *nz, uint32_t opitch, uint32_t nt_mask, hipStream_t s)
{
    const bool nts = (nt_mask & 0x100u) != 0;  // the height plane is re-read by neighbouring rows: never marked
    const uint64_t total = (uint64_t)((w + 3) / 4) * h;
    if (total == 0) return hipSuccess;
    const uint32_t row_units = (w + 3) / 4;
    // Tile width: 128, 64 or 32 quads, the widest that cuts the row into a multiple of 8 column blocks (widths that are
    // multiples of 4096, 2048 or 1024 pixels), else the widest the row holds -- rows shared inside the workgroup pay even when
    // the column blocks wander over the XCDs (3000^2: 29.5 -> 26.3 us).  KC_H2N_TILED=0: the plain mapping (A/B).
    static const int tiled_env = std::getenv("KC_H2N_TILED") ? std::atoi(std::getenv("KC_H2N_TILED")) : -1;
    uint32_t tq = 0;
    for (uint32_t t : { 7u, 6u, 5u })
        if (!tq && row_units % (8u << t) == 0) tq = t;
    for (uint32_t t : { 7u, 6u, 5u })
        if (!tq && row_units >= (1u << t)) tq = t;
    bool tiled = tiled_env != 0 && tq != 0;
    if (!tq) tq = 7;
    if ((((uint64_t)h + (256u >> tq) - 1) >> (8u - tq)) >= 65536u) tiled = false;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > grid_cap(1u << 30)) blocks = grid_cap(1u << 30);
    const dim3 grid = tiled ? dim3((row_units + (1u << tq) - 1u) >> tq, (h + (256u >> tq) - 1u) >> (8u - tq)) : dim3((unsigned)blocks);
    if (!band) full_h = h;
#define KC_H2N(BAND, NT)                                                                                                         \
    do {                                                                                                                         \
        if (tiled) height_to_normal_kernel<BAND, NT, true><<<grid, 256, 0, s>>>(hgt, hpitch, w, h, full_h, nx, ny, nz, opitch, tq);   \
        else height_to_normal_kernel<BAND, NT, false><<<grid, 256, 0, s>>>(hgt, hpitch, w, h, full_h, nx, ny, nz, opitch, tq);        \
    } while (0)
    if (band && nts) KC_H2N(true, true);
    else if (band) KC_H2N(true, false);
    else if (nts) KC_H2N(false, true);
    else KC_H2N(false, false);
#undef KC_H2N
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// u8 boundary.  to_u8 / to_u8_srgb: src/slot_image.rs:141-207, srgb_to_linear:
// src/slot_data.rs:100-109.  ((v.clamp(0,1) * 255.).min(255.)) as u8: truncation, NaN -> 255.
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ uint32_t quant_u8(float v)
{
    float x = v;
    if (x < 0.0f) x = 0.0f;
    if (x > 1.0f) x = 1.0f;  // NaN falls through both
    x = x * 255.0f;
    if (!(x <= 255.0f)) x = 255.0f;  // f32::min(255.): NaN -> 255
    return (uint32_t)x;               // 0 <= x <= 255: truncation
}

// to_u8_srgb as a step function.  q(x) = ((srgb_to_linear(x.clamp(0, 1)) * 255.).min(255.)) as u8 is non-decreasing in x, so
// q(x) = #{v : x >= T[v]} with T[v] the smallest float that exports as >= v.  The table (srgb_thresholds.inc) is generated
// with libm's powf -- what the reference's f32::powf calls -- and the identity is checked there for every float in [0, 1]
// (tools/gen_srgb_thresholds.c), so this form returns exactly what the reference's power does, without computing one: a
// hardware log2 / exp2 estimate lands within a level of the answer and two table comparisons settle it.
#include "srgb_thresholds.inc"

static __device__ __forceinline__ uint32_t quant_u8_srgb(float v, const uint32_t *T)
{
    float x = v;
    if (x < 0.0f) x = 0.0f;
    if (x > 1.0f) x = 1.0f;
    if (x != x) return 255u;  // NaN survives the clamp and the power; f32::min(255.) then returns 255
    const uint32_t xb = __float_as_uint(x);  // non-negative floats order like their bit patterns
    if ((int32_t)xb <= 0) return 0u;         // +0.0, and -0.0 (which passes the clamp): srgb_to_linear returns s itself
    const float est = 255.0f * __builtin_amdgcn_exp2f(2.4f * __builtin_amdgcn_logf((x + 0.055f) * (1.0f / 1.055f)));
    // The estimate is within one level of the answer for every float in [0, 1] (checked exhaustively on the device by
    // profiles/srgb_exhaustive.py: all 1 065 353 217 of them through this kernel against the table's definition), so one
    // comparison each way settles it: no data-dependent loop.  T[0] = 0 and the sentinel T[256] = 0xffffffff keep the
    // look-ups inside the table at both ends.
    const uint32_t q = xb < T[1] ? 0u : (uint32_t)fminf(est, 255.0f);
    return q + (xb >= T[q + 1u] ? 1u : 0u) - (xb < T[q] ? 1u : 0u);
}

template <bool NT>
static __device__ __forceinline__ float4 load_operand4(const Operand &o, uint32_t row, uint32_t q)
{
    if (o.ptr == nullptr) return splat4(o.c);
    const f4 v = ld_policy<NT>(reinterpret_cast<const f4 *>(o.ptr + (size_t)row * o.pitch + 4 * q));
    return make_float4(v.x, v.y, v.z, v.w);
}

template <bool SRGB, bool NT>  // NT: the planes are read once and do not fit the Infinity Cache (cache_policy_mask)
__global__ __launch_bounds__(256) void to_u8_kernel(Operand r, Operand g, Operand b, Operand a, int gray, uint32_t w,
                                                    uint32_t h, uint8_t *__restrict__ dst)
{
    __shared__ uint32_t srgb_t[SRGB ? 257 : 1];
    const uint32_t *pow_tab = srgb_t;
    const uint32_t row_units = (w + 3) / 4;
    const uint32_t total = row_units * h;
    auto load4 = [&](uint32_t y, uint32_t q, float4 &vr, float4 &vg, float4 &vb, float4 &va) {
        vr = load_operand4<NT>(r, y, q);
        if (!gray) {
            vg = load_operand4<NT>(g, y, q);
            vb = load_operand4<NT>(b, y, q);
            va = load_operand4<NT>(a, y, q);
        }
    };
    auto quantise_store = [&](uint32_t y, uint32_t q, const float4 &vr, const float4 &vg, const float4 &vb, const float4 &va) {
        float rr[4] = { vr.x, vr.y, vr.z, vr.w };
        uint32_t px[4];
        if (gray) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t v = SRGB ? quant_u8_srgb(rr[e], pow_tab) : quant_u8(rr[e]);
                px[e] = v | (v << 8) | (v << 16) | (255u << 24);
            }
        } else {
            float gg[4] = { vg.x, vg.y, vg.z, vg.w };
            float bb[4] = { vb.x, vb.y, vb.z, vb.w };
            float aa[4] = { va.x, va.y, va.z, va.w };
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t qr = SRGB ? quant_u8_srgb(rr[e], pow_tab) : quant_u8(rr[e]);
                const uint32_t qg = SRGB ? quant_u8_srgb(gg[e], pow_tab) : quant_u8(gg[e]);
                const uint32_t qb = SRGB ? quant_u8_srgb(bb[e], pow_tab) : quant_u8(bb[e]);
                const uint32_t qa = quant_u8(aa[e]);
                px[e] = qr | (qg << 8) | (qb << 16) | (qa << 24);
            }
        }
        uint32_t *o = reinterpret_cast<uint32_t *>(dst) + (size_t)y * w + 4 * q;
        if (4 * q + 3 < w && (w & 3u) == 0) {
            *reinterpret_cast<uint4 *>(o) = make_uint4(px[0], px[1], px[2], px[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < w) o[e] = px[e];
        }
    };
    uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if constexpr (SRGB) {
        // The first quad's plane loads go out BEFORE the threshold table is staged (a global read and a barrier that every
        // thread of the workgroup takes, in range or not): the table arrives while they are in flight (60.1 -> 57.2 us).
        const bool in_range = idx < total;
        const uint32_t y = in_range ? idx / row_units : 0u, q = in_range ? idx - y * row_units : 0u;
        float4 vr = make_float4(0, 0, 0, 0), vg = vr, vb = vr, va = vr;
        if (in_range) load4(y, q, vr, vg, vb, va);
        srgb_t[threadIdx.x] = kSrgbThresholdBits[threadIdx.x];  // 256 threads
        if (threadIdx.x == 0) srgb_t[256] = 0xffffffffu;         // sentinel: nothing is >= it
        __syncthreads();
        if (!in_range) return;
        quantise_store(y, q, vr, vg, vb, va);
        idx += gridDim.x * 256u;
    }
    for (; idx < total; idx += gridDim.x * 256u) {
        const uint32_t y = idx / row_units;
        const uint32_t q = idx - y * row_units;
        float4 vr, vg = make_float4(0, 0, 0, 0), vb = vg, va = vg;
        load4(y, q, vr, vg, vb, va);
        quantise_store(y, q, vr, vg, vb, va);
    }
}

hipError_t launch_to_u8(Operand r, Operand g, Operand b, Operand a, int gray, int srgb, uint32_t w, uint32_t h,
                        uint8_t *dst, uint32_t nt_mask, hipStream_t s)
{
    const bool ntl = (nt_mask & 0xffu) != 0;
    const uint64_t total = (uint64_t)((w + 3) / 4) * h;
    if (total == 0) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > grid_cap(1u << 30)) blocks = grid_cap(1u << 30);
    if (srgb && ntl)
        to_u8_kernel<true, true><<<dim3((unsigned)blocks), 256, 0, s>>>(r, g, b, a, gray, w, h, dst);
    else if (srgb)
        to_u8_kernel<true, false><<<dim3((unsigned)blocks), 256, 0, s>>>(r, g, b, a, gray, w, h, dst);
    else if (ntl)
        to_u8_kernel<false, true><<<dim3((unsigned)blocks), 256, 0, s>>>(r, g, b, a, gray, w, h, dst);
    else
        to_u8_kernel<false, false><<<dim3((unsigned)blocks), 256, 0, s>>>(r, g, b, a, gray, w, h, dst);
    return hipGetLastError();
}

// deconstruct_image, src/shared.rs:16-56: interleaved u8 (1..4 channels) -> planar f32 / 255.;
// channels the file lacks become constant planes on the host side (R,G,B = 0, A = 1).
template <bool NT>  // NT: the planes written do not fit the Infinity Cache (cache_policy_mask)
__global__ __launch_bounds__(256) void from_u8_kernel(const uint8_t *__restrict__ src, int channels, uint32_t w,
                                                      uint32_t h, float *p0, float *p1, float *p2, float *p3,
                                                      uint32_t pitch)
{
    const uint32_t total = w * h;
    float *planes[4] = { p0, p1, p2, p3 };
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const uint32_t y = idx / w;
        const uint32_t x = idx - y * w;
        if (channels == 4) {
            const uint32_t v = reinterpret_cast<const uint32_t *>(src)[idx];
            st_policy<NT>(&p0[(size_t)y * pitch + x], (float)(v & 255u) / 255.0f);
            st_policy<NT>(&p1[(size_t)y * pitch + x], (float)((v >> 8) & 255u) / 255.0f);
            st_policy<NT>(&p2[(size_t)y * pitch + x], (float)((v >> 16) & 255u) / 255.0f);
            st_policy<NT>(&p3[(size_t)y * pitch + x], (float)(v >> 24) / 255.0f);
        } else {
            for (int c = 0; c < channels; ++c)
                planes[c][(size_t)y * pitch + x] = (float)src[(size_t)idx * channels + c] / 255.0f;
        }
    }
}

hipError_t launch_from_u8(const uint8_t *src, int channels, uint32_t w, uint32_t h, float *const planes[4],
                          uint32_t pitch, uint32_t nt_mask, hipStream_t s)
{
    const uint64_t total = (uint64_t)w * h;
    if (total == 0) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > grid_cap(1u << 30)) blocks = grid_cap(1u << 30);
    if (nt_mask & 0x100u)
        from_u8_kernel<true><<<dim3((unsigned)blocks), 256, 0, s>>>(src, channels, w, h, planes[0], planes[1], planes[2], planes[3], pitch);
    else
        from_u8_kernel<false><<<dim3((unsigned)blocks), 256, 0, s>>>(src, channels, w, h, planes[0], planes[1], planes[2], planes[3], pitch);
    return hipGetLastError();
}

}  // namespace kc
