/*
 * kc_oracle.c -- CPU restatement of the kanter_core / vismut_core 0.10.0 per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kanter_core_amd/ may call, link or import this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity status: PINNED at u8 against the reference's own golden PNGs (tests/golden/test_compare,
 * 21 images, see tests/test_oracle_golden.py) for Mix x5 (gray/rgba), as_type, from_value,
 * Separate/Combine, Triangle up-sampling (110->128), 1x1 broadcast resize, HeightToNormal,
 * deconstruct_image and to_u8.  UNPINNED (no reference fixture exists): Nearest /
 * CatmullRom / Gaussian / Lanczos3 resampling and every down-sample; those follow the published
 * algorithm of crate `image` 0.24.0 (imageops/sample.rs), which is not vendored under /root/reference.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * Arithmetic is plain IEEE f32, no FMA contraction (build with -ffp-contract=off), libm for
 * powf / sinf / expf exactly as Rust's f32::{powf,sin,exp} lower to on Linux.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define KCO_API __attribute__((visibility("default")))

/* MixType discriminants, src/node/mix.rs:20-27 */
enum { KCO_ADD = 0, KCO_SUBTRACT = 1, KCO_MULTIPLY = 2, KCO_DIVIDE = 3, KCO_POW = 4 };
/* ResizeFilter, src/node/mod.rs:62-69 */
enum { KCO_NEAREST = 0, KCO_TRIANGLE = 1, KCO_CATMULLROM = 2, KCO_GAUSSIAN = 3, KCO_LANCZOS3 = 4 };
/* ResizePolicy, src/node/mod.rs:33-41 */
enum {
    KCO_MOST_PIXELS = 0, KCO_LEAST_PIXELS = 1, KCO_LARGEST_AXES = 2, KCO_SMALLEST_AXES = 3,
    KCO_SPECIFIC_SLOT = 4, KCO_SPECIFIC_SIZE = 5
};

static int g_threads = 1;

/* Number of OpenMP threads the row-parallel loops may use (1 = the reference's one thread per
 * node, src/engine.rs:288).  Returns the value in effect. */
KCO_API int kco_set_threads(int n)
{
#ifdef _OPENMP
    if (n < 1) n = 1;
    g_threads = n;
#else
    (void)n;
    g_threads = 1;
#endif
    return g_threads;
}

KCO_API int kco_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------------------------------------
 * Mix: src/node/mix.rs:136-192 (process_{add,subtract,multiply,divide,pow}_gray).  One plane;
 * the RGBA forms (mix.rs:194-302) call this on R, G, B and fill A with 1.0 (kco_fill).
 * ------------------------------------------------------------------------------------------- */
KCO_API int kco_mix_plane(int op, const float *l, const float *r, float *out, size_t n)
{
    long i, nn = (long)n;
    switch (op) {
    case KCO_ADD:
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (i = 0; i < nn; i++) out[i] = l[i] + r[i];
        break;
    case KCO_SUBTRACT:
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (i = 0; i < nn; i++) out[i] = l[i] - r[i];
        break;
    case KCO_MULTIPLY:
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (i = 0; i < nn; i++) out[i] = l[i] * r[i];
        break;
    case KCO_DIVIDE:
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (i = 0; i < nn; i++) out[i] = l[i] / r[i];
        break;
    case KCO_POW:
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (i = 0; i < nn; i++) out[i] = powf(l[i], r[i]);
        break;
    default:
        return -1;
    }
    return 0;
}

/* vec![value; n]: src/slot_image.rs:28-64 (from_value), src/node/mix.rs:203-211 (alpha = 1.0),
 * src/node/combine_rgba.rs:45-60 (default planes). */
KCO_API void kco_fill(float *out, size_t n, float v)
{
    long i, nn = (long)n;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (i = 0; i < nn; i++) out[i] = v;
}

/* SlotImage::as_type, RGBA -> Gray: src/slot_image.rs:242-253  ((r + g + b) / 3.) */
KCO_API void kco_rgba_to_gray(const float *r, const float *g, const float *b, float *out, size_t n)
{
    long i, nn = (long)n;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (i = 0; i < nn; i++) out[i] = ((r[i] + g[i]) + b[i]) / 3.0f;
}

/* ---------------------------------------------------------------------------------------------
 * Resize: call sites src/shared.rs:159-199 -> image::imageops::resize (crate image 0.24.0,
 * imageops/sample.rs; source NOT under /root/reference -- algorithm restated from the published
 * crate).  Vertical pass first into an unclamped f32 intermediate
 * (src_w x dst_h), then horizontal pass whose result is clamped to [0, 1].
 * ------------------------------------------------------------------------------------------- */
static float k_sinc(float t)
{
    float a = t * 3.14159265358979323846f;
    if (t == 0.0f) return 1.0f;
    return sinf(a) / a;
}

static float k_lanczos3(float x)
{
    if (fabsf(x) < 3.0f) return k_sinc(x) * k_sinc(x / 3.0f);
    return 0.0f;
}

static float k_catmullrom(float x)
{
    /* bc_cubic_spline(x, b = 0.0, c = 0.5); powi(3) = (a*a)*a, powi(2) = a*a */
    const float b = 0.0f, c = 0.5f;
    float a = fabsf(x);
    float k;
    if (a < 1.0f) {
        k = (12.0f - 9.0f * b - 6.0f * c) * ((a * a) * a) + (-18.0f + 12.0f * b + 6.0f * c) * (a * a)
            + (6.0f - 2.0f * b);
    } else if (a < 2.0f) {
        k = (-b - 6.0f * c) * ((a * a) * a) + (6.0f * b + 30.0f * c) * (a * a)
            + (-12.0f * b - 48.0f * c) * a + (8.0f * b + 24.0f * c);
    } else {
        k = 0.0f;
    }
    return k / 6.0f;
}

static float k_gaussian(float x)
{
    /* gaussian(x, r = 0.5) */
    const float r = 0.5f;
    float norm = 1.0f / (sqrtf(2.0f * 3.14159265358979323846f) * r);
    return norm * expf(-(x * x) / (2.0f * (r * r)));
}

static float k_triangle(float x)
{
    if (fabsf(x) < 1.0f) return 1.0f - fabsf(x);
    return 0.0f;
}

static float k_box(float x)
{
    (void)x;
    return 1.0f;
}

typedef float (*kernel_fn)(float);

static int filter_lookup(int filter, kernel_fn *k, float *support)
{
    switch (filter) {
    case KCO_NEAREST: *k = k_box; *support = 0.0f; return 0;
    case KCO_TRIANGLE: *k = k_triangle; *support = 1.0f; return 0;
    case KCO_CATMULLROM: *k = k_catmullrom; *support = 2.0f; return 0;
    case KCO_GAUSSIAN: *k = k_gaussian; *support = 3.0f; return 0;
    case KCO_LANCZOS3: *k = k_lanczos3; *support = 3.0f; return 0;
    }
    return -1;
}

static int64_t clamp_i64(int64_t a, int64_t lo, int64_t hi)
{
    if (a < lo) return lo;
    if (a > hi) return hi;
    return a;
}

/* image::math::utils::clamp -- NaN passes through (neither comparison holds). */
static float clamp_f32(float a, float lo, float hi)
{
    if (a < lo) return lo;
    if (a > hi) return hi;
    return a;
}

/* Per-axis tap table of one {vertical,horizontal}_sample loop: for output index o the source
 * window [left[o], left[o] + count[o]) and its normalised weights w[o * stride + j].
 * Returns the stride (max taps) or -1.  Caller frees *left, *count, *w. */
static int build_taps(uint32_t in_n, uint32_t out_n, int filter, uint32_t **left_o, uint32_t **count_o,
                      float **w_o)
{
    kernel_fn kern;
    float support;
    if (filter_lookup(filter, &kern, &support) != 0 || in_n == 0 || out_n == 0) return -1;

    float ratio = (float)in_n / (float)out_n;
    float sratio = ratio < 1.0f ? 1.0f : ratio;
    float src_support = support * sratio;

    uint32_t *left = (uint32_t *)malloc(sizeof(uint32_t) * out_n);
    uint32_t *count = (uint32_t *)malloc(sizeof(uint32_t) * out_n);
    uint32_t stride = 1;
    for (uint32_t o = 0; o < out_n; o++) {
        float input = ((float)o + 0.5f) * ratio;
        int64_t l = (int64_t)floorf(input - src_support);
        l = clamp_i64(l, 0, (int64_t)in_n - 1);
        int64_t r = (int64_t)ceilf(input + src_support);
        r = clamp_i64(r, l + 1, (int64_t)in_n);
        left[o] = (uint32_t)l;
        count[o] = (uint32_t)(r - l);
        if (count[o] > stride) stride = count[o];
    }
    float *w = (float *)calloc((size_t)out_n * stride, sizeof(float));
    for (uint32_t o = 0; o < out_n; o++) {
        float input = ((float)o + 0.5f) * ratio;
        input = input - 0.5f;
        float sum = 0.0f;
        float *wo = w + (size_t)o * stride;
        for (uint32_t j = 0; j < count[o]; j++) {
            float wi = kern(((float)(left[o] + j) - input) / sratio);
            wo[j] = wi;
            sum += wi;
        }
        for (uint32_t j = 0; j < count[o]; j++) wo[j] /= sum;
    }
    *left_o = left;
    *count_o = count;
    *w_o = w;
    return (int)stride;
}

/* Exposes the tap table so tests can compare the product's host-built tables with it. */
KCO_API int kco_resize_taps(uint32_t in_n, uint32_t out_n, int filter, uint32_t *left, uint32_t *count,
                            float *w, uint32_t w_stride)
{
    uint32_t *l, *c;
    float *ww;
    int stride = build_taps(in_n, out_n, filter, &l, &c, &ww);
    if (stride < 0) return -1;
    if (left && count && w) {
        if ((uint32_t)stride > w_stride) {
            free(l); free(c); free(ww);
            return -2;
        }
        for (uint32_t o = 0; o < out_n; o++) {
            left[o] = l[o];
            count[o] = c[o];
            for (uint32_t j = 0; j < w_stride; j++) w[(size_t)o * w_stride + j] = j < c[o] ? ww[(size_t)o * stride + j] : 0.0f;
        }
    }
    free(l); free(c); free(ww);
    return stride;
}

KCO_API int kco_resize_plane(const float *src, uint32_t sw, uint32_t sh, float *dst, uint32_t dw, uint32_t dh,
                             int filter)
{
    uint32_t *vl, *vc, *hl, *hc;
    float *vw, *hw;
    int vs = build_taps(sh, dh, filter, &vl, &vc, &vw);
    if (vs < 0) return -1;
    int hs = build_taps(sw, dw, filter, &hl, &hc, &hw);
    if (hs < 0) {
        free(vl); free(vc); free(vw);
        return -1;
    }
    /* vertical_sample: out (sw x dh), unclamped f32 */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)sw * dh);
    long oy, nrow = (long)dh;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (oy = 0; oy < nrow; oy++) {
        const float *wo = vw + (size_t)oy * vs;
        for (uint32_t x = 0; x < sw; x++) {
            float t = 0.0f;
            for (uint32_t j = 0; j < vc[oy]; j++) t += src[(size_t)(vl[oy] + j) * sw + x] * wo[j];
            tmp[(size_t)oy * sw + x] = t;
        }
    }
    /* horizontal_sample on the intermediate, clamped to [DEFAULT_MIN_VALUE, DEFAULT_MAX_VALUE] = [0, 1] */
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (oy = 0; oy < nrow; oy++) {
        for (uint32_t ox = 0; ox < dw; ox++) {
            const float *wo = hw + (size_t)ox * hs;
            float t = 0.0f;
            for (uint32_t j = 0; j < hc[ox]; j++) t += tmp[(size_t)oy * sw + hl[ox] + j] * wo[j];
            dst[(size_t)oy * dw + ox] = clamp_f32(t, 0.0f, 1.0f);
        }
    }
    free(tmp);
    free(vl); free(vc); free(vw);
    free(hl); free(hc); free(hw);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * calculate_size: src/shared.rs:61-139.  sizes[] are the inputs in the engine's edge insertion
 * order (src/engine.rs:213-218,261-275).  For SpecificSlot the caller passes slot_index = index
 * into sizes[] of the input the policy resolves to (shared.rs:113-131), or -1 when no edge.
 * ------------------------------------------------------------------------------------------- */
KCO_API int kco_calculate_size(int policy, const uint32_t *widths, const uint32_t *heights, int n, int slot_index,
                               uint32_t spec_w, uint32_t spec_h, uint32_t *out_w, uint32_t *out_h)
{
    int i, best;
    switch (policy) {
    case KCO_MOST_PIXELS:
        if (n == 0) { *out_w = 1; *out_h = 1; return 0; }
        best = 0; /* Iterator::max_by keeps the LAST maximum */
        for (i = 1; i < n; i++)
            if ((uint32_t)(widths[i] * heights[i]) >= (uint32_t)(widths[best] * heights[best])) best = i;
        *out_w = widths[best]; *out_h = heights[best];
        return 0;
    case KCO_LEAST_PIXELS:
        if (n == 0) return -1; /* unwrap() on None panics in the reference */
        best = 0; /* Iterator::min_by keeps the FIRST minimum */
        for (i = 1; i < n; i++)
            if ((uint32_t)(widths[i] * heights[i]) < (uint32_t)(widths[best] * heights[best])) best = i;
        *out_w = widths[best]; *out_h = heights[best];
        return 0;
    case KCO_LARGEST_AXES: {
        uint32_t w = 0, h = 0;
        for (i = 0; i < n; i++) { if (widths[i] > w) w = widths[i]; if (heights[i] > h) h = heights[i]; }
        *out_w = w; *out_h = h;
        return 0;
    }
    case KCO_SMALLEST_AXES: {
        uint32_t w = UINT32_MAX, h = UINT32_MAX;
        for (i = 0; i < n; i++) { if (widths[i] < w) w = widths[i]; if (heights[i] < h) h = heights[i]; }
        *out_w = w; *out_h = h;
        return 0;
    }
    case KCO_SPECIFIC_SLOT:
        if (slot_index >= 0 && slot_index < n) { *out_w = widths[slot_index]; *out_h = heights[slot_index]; }
        else { *out_w = 1; *out_h = 1; }
        return 0;
    case KCO_SPECIFIC_SIZE:
        *out_w = spec_w; *out_h = spec_h;
        return 0;
    }
    return -1;
}

/* ---------------------------------------------------------------------------------------------
 * HeightToNormal: src/node/height_to_normal.rs:16-77, wrap helpers src/node/process_shared.rs:31-65,
 * nalgebra 0.29 Vector3::{normalize, cross}: norm = sqrt((x*x + y*y) + z*z), normalize = v / norm.
 * ------------------------------------------------------------------------------------------- */
static void vnorm3(float x, float y, float z, float *o)
{
    float n = sqrtf((x * x + y * y) + z * z);
    o[0] = x / n; o[1] = y / n; o[2] = z / n;
}

KCO_API void kco_height_to_normal(const float *hgt, uint32_t w, uint32_t h, float *nx, float *ny, float *nz)
{
    float pdx = 1.0f / (float)w;
    float pdy = 1.0f / (float)h;
    long y, hh = (long)h;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (y = 0; y < hh; y++) {
        uint32_t yu = y == 0 ? h - 1 : (uint32_t)y - 1;
        for (uint32_t x = 0; x < w; x++) {
            uint32_t xl = x == 0 ? w - 1 : x - 1;
            float px = hgt[(size_t)y * w + x];
            float up = hgt[(size_t)yu * w + x];
            float left = hgt[(size_t)y * w + xl];
            float t[3], b[3], c[3], n[3];
            vnorm3(pdx, 0.0f, px - left, t);
            vnorm3(0.0f, pdy, up - px, b);
            c[0] = t[1] * b[2] - t[2] * b[1];
            c[1] = t[2] * b[0] - t[0] * b[2];
            c[2] = t[0] * b[1] - t[1] * b[0];
            vnorm3(c[0], c[1], c[2], n);
            nx[(size_t)y * w + x] = n[0] * 0.5f + 0.5f;
            ny[(size_t)y * w + x] = n[1] * 0.5f + 0.5f;
            nz[(size_t)y * w + x] = n[2] * 0.5f + 0.5f;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * u8 boundary.  deconstruct_image: src/shared.rs:16-56 (interleaved u8 -> planar f32 / 255.,
 * missing R,G,B -> 0, A -> 1).  to_u8: src/slot_image.rs:141-170; to_u8_srgb: :172-207;
 * srgb_to_linear: src/slot_data.rs:100-109.
 * ------------------------------------------------------------------------------------------- */
KCO_API void kco_deconstruct_u8(const uint8_t *px, size_t pixel_count, int channel_count, float *r, float *g,
                                float *b, float *a)
{
    float *planes[4] = { r, g, b, a };
    long i, nn = (long)pixel_count;
    for (int c = 0; c < 4; c++) {
        float *p = planes[c];
        if (c < channel_count) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
            for (i = 0; i < nn; i++) p[i] = (float)px[(size_t)i * channel_count + c] / 255.0f;
        } else {
            kco_fill(p, pixel_count, c == 3 ? 1.0f : 0.0f);
        }
    }
}

/* ((value.clamp(0.0, 1.0) * 255.).min(255.)) as u8 -- Rust `as u8` saturates and maps NaN to 0,
 * but f32::min drops the NaN first, so NaN -> 255. */
static uint8_t f32_to_u8(float v)
{
    float x = v;
    if (x < 0.0f) x = 0.0f;
    if (x > 1.0f) x = 1.0f;
    x = x * 255.0f;
    x = fminf(x, 255.0f);
    if (x != x) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}

static float srgb_to_linear(float s)
{
    if (s <= 0.0f) return s;
    if (s <= 0.04045f) return s / 12.92f;
    return powf((s + 0.055f) / 1.055f, 2.4f);
}

static uint8_t f32_to_u8_srgb(float v)
{
    float x = v;
    if (x < 0.0f) x = 0.0f;
    if (x > 1.0f) x = 1.0f;
    x = srgb_to_linear(x) * 255.0f;
    x = fminf(x, 255.0f);
    if (x != x) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}

KCO_API void kco_to_u8_rgba(const float *r, const float *g, const float *b, const float *a, size_t n, int srgb,
                            uint8_t *out)
{
    long i, nn = (long)n;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (i = 0; i < nn; i++) {
        if (srgb) {
            out[4 * i + 0] = f32_to_u8_srgb(r[i]);
            out[4 * i + 1] = f32_to_u8_srgb(g[i]);
            out[4 * i + 2] = f32_to_u8_srgb(b[i]);
        } else {
            out[4 * i + 0] = f32_to_u8(r[i]);
            out[4 * i + 1] = f32_to_u8(g[i]);
            out[4 * i + 2] = f32_to_u8(b[i]);
        }
        out[4 * i + 3] = f32_to_u8(a[i]);
    }
}

KCO_API void kco_to_u8_gray(const float *v, size_t n, int srgb, uint8_t *out)
{
    long i, nn = (long)n;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (i = 0; i < nn; i++) {
        uint8_t q = srgb ? f32_to_u8_srgb(v[i]) : f32_to_u8(v[i]);
        out[4 * i + 0] = q;
        out[4 * i + 1] = q;
        out[4 * i + 2] = q;
        out[4 * i + 3] = 255;
    }
}

/* ---------------------------------------------------------------------------------------------
 * CPU baseline for bench.py (kind "port"): the 32-node linear graph of SURVEY.md 8(d) config #3
 * evaluated the way the reference does it -- one node at a time (src/engine.rs:288: one thread
 * per node, a linear chain is serial), R, G, B planes sequentially then a freshly allocated
 * alpha plane of 1.0 (src/node/mix.rs:199-213), every node output a new allocation, the 1x1
 * Value(1.0) left operand of each invert resized to a full plane per consuming node
 * (src/shared.rs:152-207, clamp(v) broadcast) before Mix(Subtract) reads it.
 * a/b: 3 planes each (R,G,B; Mix ignores input alpha), n pixels per plane.  out: 4 planes.
 * ------------------------------------------------------------------------------------------- */
/* Plane allocation of kco_chain32.  0 (default): every node output is a new allocation that is freed when the node after it has
 * run, as the reference does (fresh pages from the kernel for each 64 MiB plane, first touched inside the node's loop).
 * 1: planes released by a node are kept on a list and handed to the next node that needs one -- what a many-core run needs to
 * scale at all (the page faults of fresh mappings serialise in the kernel); the arithmetic is the same. */
static int g_plane_pool = 0;
KCO_API int kco_set_plane_pool(int on)
{
    g_plane_pool = on != 0;
    return g_plane_pool;
}

#define KCO_POOL_MAX 16
typedef struct {
    float *free_list[KCO_POOL_MAX];
    int n_free;
    size_t n;
} kco_pool;

static float *pool_get(kco_pool *p)
{
    if (g_plane_pool && p->n_free > 0) return p->free_list[--p->n_free];
    return (float *)malloc(sizeof(float) * p->n);
}

static void pool_put(kco_pool *p, float *q)
{
    if (!q) return;
    if (g_plane_pool && p->n_free < KCO_POOL_MAX) p->free_list[p->n_free++] = q;
    else free(q);
}

KCO_API int kco_chain32(const float *const a[3], const float *const b[3], float *const out[4], uint32_t w,
                        uint32_t h, int n_nodes)
{
    size_t n = (size_t)w * h;
    kco_pool pool;
    pool.n_free = 0;
    pool.n = n;
    float *cur[4] = { 0, 0, 0, 0 };
    const float *x[3] = { a[0], a[1], a[2] };
    for (int i = 1; i <= n_nodes; i++) {
        float *nxt[4];
        for (int c = 0; c < 4; c++) {
            nxt[c] = pool_get(&pool);
            if (!nxt[c]) return -1;
        }
        if (i & 1) {
            int op = ((i >> 1) & 1) ? KCO_MULTIPLY : KCO_ADD;
            for (int c = 0; c < 3; c++) kco_mix_plane(op, x[c], b[c], nxt[c], n);
        } else {
            /* left = CombineRgba(Value 1 x3) resized 1x1 -> w x h: three broadcast planes + alpha */
            float one = 1.0f;
            float *white[4];
            for (int c = 0; c < 4; c++) {
                white[c] = pool_get(&pool);
                if (!white[c]) return -1;
                kco_resize_plane(&one, 1, 1, white[c], w, h, KCO_TRIANGLE);
            }
            for (int c = 0; c < 3; c++) kco_mix_plane(KCO_SUBTRACT, white[c], x[c], nxt[c], n);
            for (int c = 0; c < 4; c++) pool_put(&pool, white[c]);
        }
        kco_fill(nxt[3], n, 1.0f);
        for (int c = 0; c < 4; c++) pool_put(&pool, cur[c]);
        for (int c = 0; c < 4; c++) cur[c] = nxt[c];
        for (int c = 0; c < 3; c++) x[c] = cur[c];
    }
    for (int c = 0; c < 4; c++) {
        if (cur[c]) {
            memcpy(out[c], cur[c], sizeof(float) * n);
            pool_put(&pool, cur[c]);
        }
    }
    while (pool.n_free > 0) free(pool.free_list[--pool.n_free]);
    return 0;
}
