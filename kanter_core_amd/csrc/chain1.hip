// One-step chain programs ahead of time: a single Mix node (src/node/mix.rs:136-192) -- what the per-node drop-in route
// (kc_mix_process per node, use_cache), a first sighting of a graph and every host without libhiprtc run -- as plain
// straight-line kernels, one per step code: one float4 per lane, 256-thread workgroups, no step table, no loop; the shape the
// run-time specialiser emits (csrc/specialize.cpp), so a single Mix costs what its specialised kernel would
// (round 2: 0.70 of the HBM peak through the interpreter for Add, 0.52 for Pow).
// Start and operand may each be a plane or a broadcast constant (a wave-uniform choice at the load); the cache policy of the
// launch (chain_program.h, nt_mask) is a compile-time property of the instantiation, as in the generated kernels.
#include "kc_internal.hpp"

namespace kc {
namespace {
#include "chain_apply.inc"
}

template <int CODE, bool NTA, bool NTB, bool NTS>  // nontemporal: the start plane, the operand plane, the result
__global__ __launch_bounds__(256) void chain1_kernel(const Chain1Args A)
{
    constexpr bool POW = CODE == CH_POW_L || CODE == CH_POW_R;
    __shared__ double pow_lds[POW ? KC_POW_TABLE_DOUBLES : 1];
    PowCtx pw{};
    if constexpr (POW) pw = pow_setup(pow_lds);  // before any thread can leave
    const uint32_t b = blockIdx.y;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= A.rows * A.row_units) return;
    uint32_t row = 0, col = idx;
    if (A.rows != 1u) {
        row = idx / A.row_units;
        col = idx - row * A.row_units;
    }
    const float sc = A.start_c[b], xc = A.operand_c[b];
    f4 acc = { sc, sc, sc, sc }, x = { xc, xc, xc, xc };
    if (A.start[b]) {
        const f4 *p = reinterpret_cast<const f4 *>(A.start[b]) + (size_t)row * A.start_pitch[b] + col;
        if constexpr (NTA) acc = __builtin_nontemporal_load(p);
        else acc = *p;
    }
    if (A.operand[b]) {
        const f4 *p = reinterpret_cast<const f4 *>(A.operand[b]) + (size_t)row * A.operand_pitch[b] + col;
        if constexpr (NTB) x = __builtin_nontemporal_load(p);
        else x = *p;
    }
    const float c = A.c[b];
    f4 r;
    r.x = apply1<CODE>(acc.x, x.x, c, &pw);
    r.y = apply1<CODE>(acc.y, x.y, c, &pw);
    r.z = apply1<CODE>(acc.z, x.z, c, &pw);
    r.w = apply1<CODE>(acc.w, x.w, c, &pw);
    f4 *o = reinterpret_cast<f4 *>(A.out[b]) + (size_t)row * A.out_pitch[b] + col;
    if (NTS) __builtin_nontemporal_store(r, o);
    else *o = r;
}

template <int CODE>
static void launch_chain1_code(const Chain1Args &a, dim3 grid, unsigned nt, hipStream_t s)
{
    switch (nt & 7u) {  // bit 0 start, bit 1 operand, bit 2 result
    case 0: chain1_kernel<CODE, false, false, false><<<grid, 256, 0, s>>>(a); break;
    case 1: chain1_kernel<CODE, true, false, false><<<grid, 256, 0, s>>>(a); break;
    case 2: chain1_kernel<CODE, false, true, false><<<grid, 256, 0, s>>>(a); break;
    case 3: chain1_kernel<CODE, true, true, false><<<grid, 256, 0, s>>>(a); break;
    case 4: chain1_kernel<CODE, false, false, true><<<grid, 256, 0, s>>>(a); break;
    case 5: chain1_kernel<CODE, true, false, true><<<grid, 256, 0, s>>>(a); break;
    case 6: chain1_kernel<CODE, false, true, true><<<grid, 256, 0, s>>>(a); break;
    default: chain1_kernel<CODE, true, true, true><<<grid, 256, 0, s>>>(a); break;
    }
}

// nt: bit 0 = the start plane, bit 1 = the operand plane, bit 2 = the result carry the nontemporal hint
hipError_t launch_chain1(const Chain1Args &a, int batch, int code, unsigned nt, hipStream_t s)
{
    if (batch < 1 || batch > KC_CHAIN_MAX_BATCH) return hipErrorInvalidValue;
    const uint64_t total = (uint64_t)a.rows * a.row_units;
    if (total == 0) return hipSuccess;
    if (total > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((total + 255) / 256), batch, 1);
    switch (code) {
    case CH_ADD: launch_chain1_code<CH_ADD>(a, grid, nt, s); break;
    case CH_SUB_L: launch_chain1_code<CH_SUB_L>(a, grid, nt, s); break;
    case CH_SUB_R: launch_chain1_code<CH_SUB_R>(a, grid, nt, s); break;
    case CH_MUL: launch_chain1_code<CH_MUL>(a, grid, nt, s); break;
    case CH_DIV_L: launch_chain1_code<CH_DIV_L>(a, grid, nt, s); break;
    case CH_DIV_R: launch_chain1_code<CH_DIV_R>(a, grid, nt, s); break;
    case CH_POW_L: launch_chain1_code<CH_POW_L>(a, grid, nt, s); break;
    case CH_POW_R: launch_chain1_code<CH_POW_R>(a, grid, nt, s); break;
    case CH_ADD_INV: launch_chain1_code<CH_ADD_INV>(a, grid, nt, s); break;
    case CH_SUBL_INV: launch_chain1_code<CH_SUBL_INV>(a, grid, nt, s); break;
    case CH_SUBR_INV: launch_chain1_code<CH_SUBR_INV>(a, grid, nt, s); break;
    case CH_MUL_INV: launch_chain1_code<CH_MUL_INV>(a, grid, nt, s); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace kc
