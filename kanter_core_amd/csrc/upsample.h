// Integer-ratio up-sampling (512 -> 4096, 1024 -> 4096 ...: what texture sizes give) as the fused resample + Mix
// chain kernel and the plain resize kernel see it.  Shared, as text, by the library (through kc_internal.hpp) and by
// the run-time specialiser, which embeds this file in front of the kernels it generates: builtin types only, no
// includes, meant to be included inside namespace kc after chain_program.h.
//
// image::imageops::resize (crate image 0.24.0, called from src/shared.rs:159-199) gives output index o of an axis the
// window left = clamp(floor(c - S), 0, in - 1) .. right = clamp(ceil(c + S), left + 1, in) with c = (o + 0.5) in / out.
// When out = R in (R a whole number) c is never a whole number, so away from the border the window is the
// taps = 2 S + 1 source samples starting at u(o) = o / R - S, and its weights depend on o mod R only.  The host checks
// exactly that, bit for bit, on the tap table it built for the general kernels (resize.cpp, up_axis_build):
//   * every window equals [u(o), u(o) + taps) cut to [0, in);
//   * outputs b_lo .. out - b_hi - 1 carry the weights of their phase o mod R; the first b_lo and last b_hi outputs
//     (windows cut at the border, weights renormalised) carry rows of their own.
// A kernel then needs no per-output table look-ups: a window start is one division, a weight row is one of
// R + b_lo + b_hi rows of `taps` floats ("classes"), slots of a window that fall outside [0, in) hold the weight
// 0.0 and read a sample 0.0 -- a product of +0.0, which leaves every partial sum of the reference's shorter
// sequence unchanged (a sum that starts at +0.0 is never -0.0) -- so all threads run the same straight code.
// With R a multiple of 4 the 4 output columns a thread owns share one window; their weights are then kept as
// "quad classes": R / 4 + qb_lo + qb_hi blocks of taps x 4 floats, tap-major, so that one 16-byte read yields tap j's
// weights for all four columns.  R = 2 ("half quads", out a multiple of 4): columns 0, 1 of a quad share the window u and
// columns 2, 3 the window u + 1; there is ONE interior quad class (phases 0, 1, 0, 1), every block in its columns' own frames.
struct UpAxis {
    unsigned int n_in, n_out;
    unsigned int ratio;  // R = n_out / n_in (< 65536)
    unsigned int magic;  // ceil(2^32 / R): o / R == umulhi(o, magic) for o < 65536
    unsigned int taps;   // window length away from the border (odd)
    int off;             // S: window of output o starts at (int)(o / ratio) - off
    unsigned int b_lo, b_hi;  // outputs with rows of their own at either end
    unsigned int qb_lo, qb_hi;  // the same in column quads (R % 4 == 0 or R == 2 only): ceil(b / 4)
    const float *cls;    // class rows (HBM): ratio + b_lo + b_hi rows of `taps` floats
    const float *qcls;   // quad classes (HBM; R % 4 == 0 or R == 2 only): max(ratio / 4, 1) + qb_lo + qb_hi blocks of taps x 4 floats
};

struct UpsampleArgs {
    UpAxis H, V;
    unsigned int tile_w;  // output columns of one workgroup; tile_w / 4 divides 256.  Its rows: rows_per_thread * 1024 / tile_w
    unsigned int ncp;     // LDS pitch of the vertical-pass intermediate in floats (a multiple of 4 covering any tile's window)
    unsigned int chunk;   // rows per vertical-pass work item: divides V.ratio and the tile's rows, so a chunk's rows share one window
};
