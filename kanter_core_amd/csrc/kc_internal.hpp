// Internal declarations shared by the HIP kernels (kernels.hip) and the host runtime.
// Not part of the C ABI (include/kanter_core_amd.h).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kanter_core_amd.h"

namespace kc {

#include "chain_program.h"  // ChainCode, ChainStepRec / ChainStepPair, ChainProgram
#include "upsample.h"       // UpAxis, UpsampleArgs

// Per-axis tap table of the separable resampler, resident in HBM.
struct TapsDev {
    const uint32_t *left;
    const uint32_t *count;
    const float *w;
    uint32_t stride;
};

// Pointwise operand: a pitched plane or a broadcast constant.
struct Operand {
    const float *ptr;  // nullptr => constant
    uint32_t pitch;    // in floats
    float c;
};

// ---- kernel launchers (kernels.hip).  All enqueue on `s` and return hipGetLastError(). ----
// mode: 0 = {+, -, *} only, 1 = + divide, 2 = + pow
hipError_t launch_chain(const ChainProgram &p, int batch, int mode, int max_blocks, int unroll, hipStream_t s);
// A one-step program for the ahead-of-time kernels of chain1.hip: result = op(start, operand), each a plane (pointer,
// pitch in float4) or a broadcast constant (pointer null); c = the constant of the fused "c - ..." codes.
struct Chain1Args {
    const float *start[KC_CHAIN_MAX_BATCH], *operand[KC_CHAIN_MAX_BATCH];
    float *out[KC_CHAIN_MAX_BATCH];
    uint32_t start_pitch[KC_CHAIN_MAX_BATCH], operand_pitch[KC_CHAIN_MAX_BATCH], out_pitch[KC_CHAIN_MAX_BATCH];
    float start_c[KC_CHAIN_MAX_BATCH], operand_c[KC_CHAIN_MAX_BATCH], c[KC_CHAIN_MAX_BATCH];
    uint32_t row_units, rows;
};
// nt: bit 0 = the start plane, bit 1 = the operand plane, bit 2 = the result carry the nontemporal hint
hipError_t launch_chain1(const Chain1Args &a, int batch, int code, unsigned nt, hipStream_t s);
inline uint32_t chain_op_word(uint8_t code, int src) { return (uint32_t)code | ((uint32_t)(src + 1) << 8); }
hipError_t launch_fill(float *dst, uint32_t pitch_floats, uint32_t w, uint32_t h, float v, hipStream_t s);
hipError_t launch_resize_vertical(const float *src, uint32_t spitch, uint32_t sw, float *tmp, uint32_t tpitch,
                                  uint32_t dh, TapsDev v, hipStream_t s);
hipError_t launch_resize_horizontal(const float *tmp, uint32_t tpitch, float *dst, uint32_t dpitch, uint32_t dw,
                                    uint32_t dh, TapsDev h, hipStream_t s);
// Tiled single-pass resample.  ncp = LDS pitch in floats of the vertical-pass intermediate: a multiple
// of 4 that covers the widest 4-aligned source window any tile needs (from the host).  The 8 spare
// floats absorb the register-tap form's reads past a short window (discarded, see resize_out_row);
// the tile rows' vertical tap table follows.
// Windows of more than KC_RESIZE_REG_TAPS horizontal taps also keep the tile's horizontal tap table there.
constexpr uint32_t KC_RESIZE_REG_TAPS = 8;
inline size_t resize_lds_bytes(uint32_t tile_h, uint32_t ncp, uint32_t v_stride, uint32_t tile_w, uint32_t h_stride)
{
    // the wide form pads its intermediate rows (one float after every 32, see resize_wide_kernel)
    const size_t row = h_stride > KC_RESIZE_REG_TAPS ? (size_t)ncp + (ncp >> 5) + 1u : ncp;
    size_t n = (size_t)tile_h * row + 8u + 2u * tile_h + (size_t)tile_h * v_stride;
    if (h_stride > KC_RESIZE_REG_TAPS) n += 2u * tile_w + (size_t)tile_w * h_stride;
    return n * sizeof(float);
}
// resize_down_kernel / resize_poly_kernel: intermediate rows of a fixed pitch (a window of at most 256 floats, one float of
// padding after every 32), then the tile's horizontal taps at an odd pitch
#define KC_DOWN_ROW_FLOATS 265u
inline size_t resize_down_lds_bytes(uint32_t tile_h, uint32_t ncp, uint32_t tile_w, uint32_t h_stride)
{
    (void)ncp;  // <= 256 (host-checked)
    return ((size_t)tile_h * KC_DOWN_ROW_FLOATS + 2u * tile_w + (size_t)tile_w * (h_stride | 1u)) * sizeof(float);
}
// Up to 4 planes of equal size (the planes of one image) resampled by one launch, blockIdx.z = plane.
struct ResizePlanes {
    const float *src[4];
    float *dst[4];
    uint32_t spitch[4], dpitch[4];  // in floats
};
// resize_down2_kernel (down2.hip): the wave-private form of down-sampling on both axes.
//   vrec     per group of 4 output rows `nc` records of KC_DOWN2_REC dwords: [0] first source row of the chunk, [1], [2] presence
//            mask of tap (source row u, output row k) at bit 4 u + k, [3] the group's last window row (loads are clamped to
//            it), [4] chunks this group uses, [5] the same in half chunks of 8 rows, [8 + 4 u + k] the weight (resize.cpp, down2_build)
//   hw       the horizontal table's weights, rows padded to hstride (a multiple of 4) floats
//   strips   per strip of tile_w output columns: its first source column rounded down to a multiple of 4, and the width of its
//            source window in column quads (<= 64: one per lane of the vertical pass)
constexpr uint32_t KC_DOWN2_REC = 72, KC_DOWN2_MAX_CHUNKS = 4, KC_DOWN2_SLOTS = 256 + 32;
struct Down2Args {
    const uint32_t *vrec;
    const uint32_t *hleft, *hcount, *strips;
    const float *hw;
    uint32_t nc, hstride;
    uint32_t tile_w, dw, dh;
    // XCD-aware tile order (down2.hip).  The caller sets xcd_per != 0 to ask for it; the launcher fills in the rest (or clears
    // xcd_per: plain 2-D grid).
    uint32_t xcd_per, n_tiles, gy, gy_magic;
    // by_rows != 0: a workgroup's four waves are four neighbouring STRIPS of one row group and XCD k works through the k-th eighth
    // of the jobs row by row (xcd_per workgroups each; gy / gy_magic then divide by the number of strips): see resize_poly_kernel
    uint32_t by_rows;
};
// output columns per lane of the horizontal pass: its weights live in registers (at most 9 quads per lane)
inline uint32_t down2_cols_per_lane(uint32_t weight_quads) { return weight_quads <= 3 ? 3u : weight_quads == 4 ? 2u : 1u; }
hipError_t launch_resize_down2(const ResizePlanes &p, int batch, const Down2Args &a, hipStream_t s);
hipError_t launch_resize_lds(const ResizePlanes &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h,
                             uint32_t h_min_count, uint32_t tile_w, uint32_t tile_h, uint32_t ncp, hipStream_t s);
hipError_t launch_resize_down(const ResizePlanes &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h, uint32_t tile_w,
                              uint32_t tile_h, uint32_t ncp, hipStream_t s);
// Tiles in XCD order (as resize_down2_kernel's): with per != 0 the grid is one-dimensional, workgroup id % 8 is the XCD and XCD k
// works through the k-th eighth of the gx x gy tiles in column-major order (tile = (id % 8) * per + id / 8; x = tile / gy), so
// that vertically adjacent tiles, which share source rows, meet in one L2.
struct XcdOrder {
    uint32_t per, n, gy, magic;
};
inline XcdOrder xcd_order(uint32_t gx, uint32_t gy, bool want)
{
    XcdOrder o{ 0, 0, 0, 0 };
    if (!want || gy < 2) return o;
    const uint64_t n = (uint64_t)gx * gy, magic = ((1ull << 32) + gy - 1) / gy;
    // tile / gy == (tile * magic) >> 32 for every tile < n when n * (magic * gy - 2^32) < 2^32
    if (n >= (1u << 24) || n * (magic * gy - (1ull << 32)) >= (1ull << 32)) return o;
    o.per = (uint32_t)((n + 7) / 8);
    o.n = (uint32_t)n;
    o.gy = gy;
    o.magic = (uint32_t)magic;
    return o;
}
hipError_t launch_resize_poly(const ResizePlanes &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h, uint32_t tile_w,
                              uint32_t ncp, uint32_t reg_a, uint32_t reg_b, uint32_t ages, uint32_t ratio, hipStream_t s);
// The same ranges with two waves to a band's strip (8-byte lanes, a shared ring, the horizontal pass split by pixels): kernels.hip
hipError_t launch_resize_poly2(const ResizePlanes &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h, uint32_t tw,
                               uint32_t gen_tw, uint32_t gen_ncp, uint32_t reg_a, uint32_t reg_b, uint32_t ages, uint32_t ratio, bool xcd,
                               hipStream_t s);
// Fused resample + chain: input slot n_in - 1 of the program is produced by the resampler.
hipError_t launch_resize_chain(const ChainProgram &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h,
                               uint32_t tile_w, uint32_t tile_h, uint32_t ncp, hipStream_t s);
// Integer-ratio up-sampling (upsample.h).  A workgroup's tile: tile_w columns x KC_UPSAMPLE_ROWS * (1024 / tile_w) rows
// (every thread 4 columns x KC_UPSAMPLE_ROWS rows, one trip).  LDS: the tile's intermediate, then the H quad classes.
#ifdef KC_UP_RU  // tuning builds (tools/build_variant.sh)
constexpr uint32_t KC_UPSAMPLE_ROWS = KC_UP_RU;
#else
constexpr uint32_t KC_UPSAMPLE_ROWS = 4;
#endif
inline uint32_t upsample_tile_rows(const UpsampleArgs &u) { return KC_UPSAMPLE_ROWS * (1024u / u.tile_w); }
inline size_t upsample_lds_bytes(const UpsampleArgs &u)
{
    return ((size_t)upsample_tile_rows(u) * u.ncp + (size_t)(std::max(u.H.ratio >> 2, 1u) + u.H.qb_lo + u.H.qb_hi) * 4u * u.H.taps) * sizeof(float);
}
inline bool upsample_args_ok(const UpsampleArgs &u, int batch)
{
    if (batch < 1 || batch > 4) return false;
    if (u.tile_w % 4 != 0 || u.tile_w == 0 || u.tile_w > 1024 || 256u % (u.tile_w / 4) != 0) return false;
    const uint32_t tile_h = upsample_tile_rows(u);
    if (u.chunk == 0 || tile_h % u.chunk != 0 || u.V.ratio % u.chunk != 0) return false;
    if (u.H.ratio % 4 != 0 && !(u.H.ratio == 2 && u.H.n_out % 4 == 0)) return false;
    if (u.H.taps != u.V.taps || u.ncp % 4 != 0 || !u.H.qcls || !u.V.cls) return false;
    if (u.H.n_out > 65535 || u.V.n_out > 65535) return false;  // up_div
    if (u.ncp / 4 > 257 || (std::max(u.H.ratio >> 2, 1u) + u.H.qb_lo + u.H.qb_hi) * u.H.taps > 256) return false;
    return upsample_lds_bytes(u) <= 64 * 1024;
}
inline dim3 upsample_grid(const UpsampleArgs &u, int batch)
{
    const uint32_t tile_h = upsample_tile_rows(u);
    return dim3((u.H.n_out + u.tile_w - 1) / u.tile_w, (u.V.n_out + tile_h - 1) / tile_h, batch);
}
// The plane members of ChainProgram that the plain up-sampling kernel reads (K = 1: no resident inputs).
struct UpsamplePlanes {
    const float *samp_src[4];
    unsigned int samp_pitch[4];  // floats
    float *out[4];
    unsigned int out_pitch[4];  // float4 units
    const float *in[4][1];
    unsigned int in_pitch[4][1];
    unsigned int nt_mask;  // bit 8: nontemporal stores (ChainProgram::nt_mask)
};
hipError_t launch_upsample_chain(const ChainProgram &p, int batch, const UpsampleArgs &u, hipStream_t s);
hipError_t launch_upsample(const UpsamplePlanes &p, int batch, const UpsampleArgs &u, hipStream_t s);
// nt_mask: the launch's cache policy as cache_policy_mask() returns it (bits 0-7: inputs, bit 8: results)
hipError_t launch_height_to_normal(const float *hgt, uint32_t hpitch, uint32_t w, uint32_t h, uint32_t full_h, int band,
                                   float *nx, float *ny, float *nz, uint32_t opitch, uint32_t nt_mask, hipStream_t s);
hipError_t launch_to_u8(Operand r, Operand g, Operand b, Operand a, int gray, int srgb, uint32_t w, uint32_t h,
                        uint8_t *dst, uint32_t nt_mask, hipStream_t s);
hipError_t launch_from_u8(const uint8_t *src, int channels, uint32_t w, uint32_t h, float *const planes[4],
                          uint32_t pitch, uint32_t nt_mask, hipStream_t s);

}  // namespace kc
