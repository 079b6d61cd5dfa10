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

// ------------------------------------------------------------------------------------------
// Fused pointwise chain: acc = start; for each step acc = op(acc, x) or op(x, acc).
// One program drives up to KC_CHAIN_MAX_BATCH planes (the R, G, B planes of an RGBA Mix share
// ops but not operands), blockIdx.y selects the plane.
// ------------------------------------------------------------------------------------------
constexpr int KC_CHAIN_MAX_OPS = 64;
constexpr int KC_CHAIN_MAX_IN = 4;
constexpr int KC_CHAIN_MAX_BATCH = 4;

// Step codes: which side the running value sits on matters for -, / and pow.
enum ChainCode : uint8_t {
    CH_ADD = 0,    // acc + x
    CH_SUB_L = 1,  // acc - x
    CH_SUB_R = 2,  // x - acc
    CH_MUL = 3,    // acc * x
    CH_DIV_L = 4,  // acc / x
    CH_DIV_R = 5,  // x / acc
    CH_POW_L = 6,  // acc ^ x
    CH_POW_R = 7,  // x ^ acc
    CH_ADD_R = 8,  // x + acc: host-side only, canonicalised to CH_ADD before launch (same IEEE sum)
    CH_MUL_R = 9,  // x * acc: host-side only, canonicalised to CH_MUL
    // Device-side only (chain_fill): a {+, -, *} step on a plane operand x followed by an invert-style
    // step "c - acc" (Mix(Subtract)(constant, .), how every graph spells 1 - x) is ONE record and one
    // dispatch.  Both roundings happen, in order: the result is that of the two separate steps.
    CH_ADD_INV = 10,   // c - (acc + x)
    CH_SUBL_INV = 11,  // c - (acc - x)
    CH_SUBR_INV = 12,  // c - (x - acc)
    CH_MUL_INV = 13    // c - (acc * x)
};

// word: bits 0-7 ChainCode, bits 8-15 operand source (0 = the constant c, k + 1 = input plane k).
struct ChainStepRec {
    uint32_t word;
    float c;
};
struct alignas(16) ChainStepPair {  // steps 2i and 2i + 1: one 16-byte scalar load
    ChainStepRec a, b;
};

struct ChainProgram {
    uint32_t n_ops;
    uint32_t n_in;
    uint32_t row_units;  // vector units (float4 or float) per row; rows * row_units = work items
    uint32_t rows;
    int32_t start_src;  // input index, or -1: start from start_c
    const float *in[KC_CHAIN_MAX_BATCH][KC_CHAIN_MAX_IN];
    uint32_t in_pitch[KC_CHAIN_MAX_BATCH][KC_CHAIN_MAX_IN];  // in vector units
    float *out[KC_CHAIN_MAX_BATCH];
    uint32_t out_pitch[KC_CHAIN_MAX_BATCH];
    float start_c[KC_CHAIN_MAX_BATCH];
    // resize_chain_kernel only: the source plane of the resampled operand (input slot n_in - 1)
    const float *samp_src[KC_CHAIN_MAX_BATCH];
    uint32_t samp_pitch[KC_CHAIN_MAX_BATCH];  // in floats
    // One 8-byte record per step and channel, fetched two at a time by one scalar (SMEM) load; one
    // spare pair lets the loop prefetch the next pair unconditionally.
    ChainStepPair step[KC_CHAIN_MAX_BATCH][KC_CHAIN_MAX_OPS / 2 + 1];
};

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
    size_t n = (size_t)tile_h * ncp + 8u + 2u * tile_h + (size_t)tile_h * v_stride;
    if (h_stride > KC_RESIZE_REG_TAPS) n += 2u * tile_w + (size_t)tile_w * h_stride;
    return n * sizeof(float);
}
// Up to 4 planes of equal size (the planes of one image) resampled by one launch, blockIdx.z = plane.
struct ResizePlanes {
    const float *src[4];
    float *dst[4];
    uint32_t spitch[4], dpitch[4];  // in floats
};
hipError_t launch_resize_lds(const ResizePlanes &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h,
                             uint32_t h_min_count, uint32_t tile_w, uint32_t tile_h, uint32_t ncp, hipStream_t s);
// Fused resample + chain: input slot n_in - 1 of the program is produced by the resampler.
hipError_t launch_resize_chain(const ChainProgram &p, int batch, uint32_t dw, uint32_t dh, TapsDev v, TapsDev h,
                               uint32_t tile_w, uint32_t tile_h, uint32_t ncp, hipStream_t s);
hipError_t launch_height_to_normal(const float *hgt, uint32_t hpitch, uint32_t w, uint32_t h, float *nx, float *ny,
                                   float *nz, uint32_t opitch, hipStream_t s);
hipError_t launch_to_u8(Operand r, Operand g, Operand b, Operand a, int gray, int srgb, uint32_t w, uint32_t h,
                        uint8_t *dst, hipStream_t s);
hipError_t launch_from_u8(const uint8_t *src, int channels, uint32_t w, uint32_t h, float *const planes[4],
                          uint32_t pitch, hipStream_t s);

}  // namespace kc
