// The fused pointwise chain's program: what the host hands the chain kernels as their one argument.
// Shared, as text, by the library (through kc_internal.hpp) and by the run-time specialiser, which embeds
// this file in front of every kernel it generates (csrc/specialize.cpp): builtin types only, no includes,
// meant to be included inside namespace kc.
//
// Fused pointwise chain: acc = start; for each step acc = op(acc, x) or op(x, acc).
// One program drives up to KC_CHAIN_MAX_BATCH planes (the R, G, B planes of an RGBA Mix share
// ops but not operands), blockIdx.y selects the plane.
// KC_CHAIN_MAX_IN input planes per channel fit a program; the interpreter, the one-step kernels and the fused resize kernels
// handle KC_CHAIN_INTERP_IN of them: a program with more runs on its own compiled kernel only (runtime.cpp chain_launch).
enum { KC_CHAIN_MAX_OPS = 80, KC_CHAIN_MAX_IN = 16, KC_CHAIN_INTERP_IN = 4, KC_CHAIN_MAX_BATCH = 4 };
// (sizes: the argument block -- 16 x 4 pointers and pitches, 41 x 4 record pairs -- is 3.6 KB of the 4 KB a launch may pass)
// Cache-policy bit of input plane k in ChainProgram::nt_mask: bits 0-7, then 16-23 (bit 8 is the result's).
#define KC_CHAIN_NT_BIT(k) ((k) < 8 ? 1u << (k) : 1u << ((k) + 8))

// Step codes: which side the running value sits on matters for -, / and pow.
enum ChainCode : unsigned char {
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
    CH_MUL_INV = 13,   // c - (acc * x)
    // Two chains in one program (runtime.cpp, plane_mix: a Mix whose two inputs are BOTH unevaluated chains).  CH_SAVE_LOAD
    // puts the running value aside and starts the second chain from x; the step that combines the two has the saved value
    // as its operand (operand source KC_CHAIN_SRC_SAVED).  Only the kernels compiled at run time implement these: the
    // interpreter never sees such a program (the host runs the second chain on its own instead, runtime.cpp chain_launch).
    CH_SAVE_LOAD = 14  // saved[level] = acc; acc = x
};
// Operand source (word bits 8-15) KC_CHAIN_SRC_SAVED: the value CH_SAVE_LOAD put aside.  A joined chain can hold joins of its
// own: KC_CHAIN_MAX_SAVED values can be aside at once, word bits 16-17 say which one a CH_SAVE_LOAD writes / a step reads.
enum { KC_CHAIN_SRC_SAVED = 255, KC_CHAIN_MAX_SAVED = 3 };  // (255: clear of k + 1 for every input plane k)

// word: bits 0-7 ChainCode, bits 8-15 operand source (0 = the constant c, k + 1 = input plane k, KC_CHAIN_SRC_SAVED), bits 16-17
// the saved value meant (CH_SAVE_LOAD, KC_CHAIN_SRC_SAVED).
struct ChainStepRec {
    unsigned int word;
    float c;
};
struct alignas(16) ChainStepPair {  // steps 2i and 2i + 1: one 16-byte scalar load
    ChainStepRec a, b;
};

struct ChainProgram {
    unsigned int n_ops;
    unsigned int n_in;
    unsigned int row_units;  // vector units (float4 or float) per row; rows * row_units = work items
    unsigned int rows;
    int start_src;  // input index, or -1: start from start_c
    // Cache policy of this launch, chosen by the host (runtime.cpp, cache_policy_mask): KC_CHAIN_NT_BIT(k) = input plane k is read with
    // the nontemporal hint (streamed once, not worth a place in the 256 MB Infinity Cache), bit 8 = the result is stored
    // with it.  Honoured by the kernels compiled at run time and by the up-sampling kernels; a hint, never semantics.
    unsigned int nt_mask;
    const float *in[KC_CHAIN_MAX_BATCH][KC_CHAIN_MAX_IN];
    unsigned int in_pitch[KC_CHAIN_MAX_BATCH][KC_CHAIN_MAX_IN];  // in vector units
    float *out[KC_CHAIN_MAX_BATCH];
    unsigned int out_pitch[KC_CHAIN_MAX_BATCH];
    float start_c[KC_CHAIN_MAX_BATCH];
    // resize_chain_kernel only: the source plane of the resampled operand (input slot n_in - 1)
    const float *samp_src[KC_CHAIN_MAX_BATCH];
    unsigned int samp_pitch[KC_CHAIN_MAX_BATCH];  // in floats
    // One 8-byte record per step and channel, fetched two at a time by one scalar (SMEM) load; one
    // spare pair lets the loop prefetch the next pair unconditionally.
    ChainStepPair step[KC_CHAIN_MAX_BATCH][KC_CHAIN_MAX_OPS / 2 + 1];
};
