// Implicit input resize: resize_buffers (src/shared.rs:141-216) -> image::imageops::resize.
// The resampler's arithmetic lives in crate `image` 0.24.0 (imageops/sample.rs), which is not
// vendored with the reference; this file follows that crate's published algorithm:
// per axis, for output index o:  ratio = in/out (f32), sratio = max(ratio, 1), S = support*sratio,
// c = (o + 0.5)*ratio, left = clamp(floor(c - S), 0, in-1), right = clamp(ceil(c + S), left+1, in),
// w_i = K((i - (c - 0.5)) / sratio), normalised by their sequential f32 sum.  The tap tables are
// built here on the host with the same libm calls (sinf / expf) Rust's f32::sin / f32::exp
// lower to, uploaded once per (in, out, filter) and cached; the kernels only multiply-add.
#include <cmath>
#include <cstdlib>

#include "kc_runtime.hpp"

namespace kc {

static float k_box(float) { return 1.0f; }

static float k_triangle(float x)
{
    const float a = std::fabs(x);
    return a < 1.0f ? 1.0f - a : 0.0f;
}

static float k_catmullrom(float x)
{
    // Mitchell-Netravali BC-spline with b = 0, c = 0.5; a.powi(3) = (a*a)*a, a.powi(2) = a*a
    const float b = 0.0f, c = 0.5f;
    const float a = std::fabs(x);
    float k = 0.0f;
    if (a < 1.0f)
        k = (12.0f - 9.0f * b - 6.0f * c) * ((a * a) * a) + (-18.0f + 12.0f * b + 6.0f * c) * (a * a) + (6.0f - 2.0f * b);
    else if (a < 2.0f)
        k = (-b - 6.0f * c) * ((a * a) * a) + (6.0f * b + 30.0f * c) * (a * a) + (-12.0f * b - 48.0f * c) * a +
            (8.0f * b + 24.0f * c);
    return k / 6.0f;
}

static float k_gaussian(float x)
{
    const float r = 0.5f;
    const float pi = 3.14159265358979323846f;
    return (1.0f / (std::sqrt(2.0f * pi) * r)) * std::exp(-(x * x) / (2.0f * (r * r)));
}

static float sinc(float t)
{
    const float a = t * 3.14159265358979323846f;
    return t == 0.0f ? 1.0f : std::sin(a) / a;
}

static float k_lanczos3(float x) { return std::fabs(x) < 3.0f ? sinc(x) * sinc(x / 3.0f) : 0.0f; }

// Does the table have the structure upsample.h describes?  Checked on the table itself, bit for bit: nothing about the
// filter or the ratio is assumed beyond out = R x in.
static void up_axis_build(uint32_t in_n, uint32_t out_n, TapsHost &t)
{
    t.up_ok = false;
    t.up_rows.clear();
    if (out_n % in_n != 0) return;
    const uint32_t R = out_n / in_n, T = t.stride;
    if (T % 2 == 0 || T > 7) return;
    const int off = (int)(T - 1) / 2;
    std::vector<float> rows((size_t)out_n * T, 0.0f);  // every window laid out from its unclamped start
    for (uint32_t o = 0; o < out_n; ++o) {
        const int u = (int)(o / R) - off;
        const int lo = std::max(u, 0), hi = std::min(u + (int)T, (int)in_n);
        if ((int)t.left[o] != lo || (int)(t.left[o] + t.count[o]) != hi) return;
        for (uint32_t j = 0; j < t.count[o]; ++j) rows[(size_t)o * T + (uint32_t)(lo - u) + j] = t.w[(size_t)o * t.stride + j];
    }
    // the phases' rows are those of a period in the middle; whatever differs from them must sit at either end
    const uint32_t ref = (in_n / 2) * R;
    auto same = [&](uint32_t o) { return std::memcmp(&rows[(size_t)o * T], &rows[(size_t)(ref + o % R) * T], T * sizeof(float)) == 0; };
    uint32_t b_lo = 0, hi_start = out_n;
    for (uint32_t o = 0; o < ref; ++o)
        if (!same(o)) b_lo = o + 1;
    for (uint32_t o = out_n; o-- > ref + R;)
        if (!same(o)) hi_start = o;
    const uint32_t b_hi = out_n - hi_start;
    if ((size_t)(R + b_lo + b_hi) * T > 2048 || R > 65535 || out_n > 65535) return;  // the rows live in LDS; up_div
    t.up = UpAxis{};
    t.up.n_in = in_n;
    t.up.n_out = out_n;
    t.up.ratio = R;
    t.up.magic = R > 1 ? (uint32_t)((0x100000000ull + R - 1) / R) : 0u;
    t.up.taps = T;
    t.up.off = off;
    t.up.b_lo = b_lo;
    t.up.b_hi = b_hi;
    t.up_rows.assign(rows.begin() + (size_t)ref * T, rows.begin() + (size_t)(ref + R) * T);
    t.up_rows.insert(t.up_rows.end(), rows.begin(), rows.begin() + (size_t)b_lo * T);
    t.up_rows.insert(t.up_rows.end(), rows.begin() + (size_t)hi_start * T, rows.end());
    // quad classes (R % 4 == 0: the four columns of a quad share their window): tap-major blocks of 4 weights
    t.up_qrows.clear();
    if (R % 4 == 0 || (R == 2 && out_n % 4 == 0 && in_n >= 8)) {
        // (R == 2: a quad's columns 0, 1 and 2, 3 have windows one sample apart; every row sits in its own column's frame, and the
        // one interior class is the quad at `ref`: phases 0, 1, 0, 1)
        const uint32_t nqx = out_n / 4, qb_lo = (b_lo + 3) / 4, qb_hi = (b_hi + 3) / 4;
        auto block = [&](uint32_t o0) {
            for (uint32_t j = 0; j < T; ++j)
                for (uint32_t e = 0; e < 4; ++e) t.up_qrows.push_back(rows[(size_t)(o0 + e) * T + j]);
        };
        for (uint32_t c = 0; c < std::max(R / 4, 1u); ++c) block(ref + 4 * c);
        for (uint32_t q = 0; q < qb_lo; ++q) block(4 * q);
        for (uint32_t q = nqx - qb_hi; q < nqx; ++q) block(4 * q);
        t.up.qb_lo = qb_lo;
        t.up.qb_hi = qb_hi;
    }
    t.up_ok = true;
}

// The widest source-column window any tile_w-wide output tile needs, measured from its first column
// rounded down to a multiple of 4, in whole 4-column groups.
static uint32_t tile_groups(const TapsHost &h, uint32_t out_n, uint32_t tile_w)
{
    uint32_t groups = 1;
    for (uint32_t x0 = 0; x0 < out_n; x0 += tile_w) {
        const uint32_t x1 = std::min(out_n, x0 + tile_w);
        const uint32_t n = (h.left[x1 - 1] + h.count[x1 - 1] - (h.left[x0] & ~3u) + 3u) / 4u;
        if (n > groups) groups = n;
    }
    return groups;
}

// What resize_down2_kernel (down2.hip) reads instead of the plain table, for the tables of down-sampling axes (d2_want).
// Vertical use: per group of four output rows the union of their windows in chunks of 16 source rows; record (group, chunk)
// holds the weight of tap (source row u of the chunk, output row k of the group) at [8 + 4 u + k] -- +0.0 and a clear mask
// bit where row k has no tap on that source row -- so the kernel needs no per-tap look-up.  The weights are the table's own
// f32 values; ascending u is ascending tap index for every row, so the sums run in the reference's order.
// Horizontal use: rows padded to a multiple of four weights, and the strip width.
void down2_build(uint32_t out_n, TapsHost &t)
{
    t.d2_nc = t.d2_hstride = t.d2_tile_w = t.p2_tile_w = 0;
    t.d2_vrec.clear();
    t.d2_strips.clear();
    t.d2_hw.clear();
    if (!t.d2_want || out_n == 0) return;
    // resize_poly2_kernel's strips: the widest (up to 128 columns) whose source window is at most 64 quads, evened out
    for (uint32_t tw = std::min(128u, out_n); tw >= 1; --tw)
        if (tile_groups(t, out_n, tw) <= 64u) {
            const uint32_t strips = (out_n + tw - 1) / tw, even = (out_n + strips - 1) / strips;
            t.p2_tile_w = tile_groups(t, out_n, even) <= 64u ? even : tw;
            break;
        }
    const uint32_t groups = (out_n + 3u) / 4u;
    uint32_t nc = 0;
    for (uint32_t g = 0; g < groups; ++g) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0;
        for (uint32_t y = 4u * g; y < std::min(out_n, 4u * g + 4u); ++y) {
            lo = std::min(lo, t.left[y]);
            hi = std::max(hi, t.left[y] + t.count[y]);
        }
        nc = std::max(nc, (hi - lo + 15u) / 16u);
    }
    if (nc >= 1 && nc <= KC_DOWN2_MAX_CHUNKS) {
        t.d2_nc = nc;
        t.d2_vrec.assign((size_t)groups * nc * KC_DOWN2_REC, 0u);
        for (uint32_t g = 0; g < groups; ++g) {
            const uint32_t y0 = 4u * g, y1 = std::min(out_n, y0 + 4u);
            uint32_t lo = 0xFFFFFFFFu, hi = 0;
            for (uint32_t y = y0; y < y1; ++y) {
                lo = std::min(lo, t.left[y]);
                hi = std::max(hi, t.left[y] + t.count[y]);
            }
            const uint32_t used = (hi - lo + 15u) / 16u;
            for (uint32_t ch = 0; ch < nc; ++ch) {
                uint32_t *r = &t.d2_vrec[((size_t)g * nc + ch) * KC_DOWN2_REC];
                const uint32_t s0 = std::min(lo + 16u * ch, hi - 1u);
                uint64_t mask = 0;
                if (ch < used)
                    for (uint32_t u = 0; u < 16u; ++u)
                        for (uint32_t y = y0; y < y1; ++y) {
                            const uint32_t s = s0 + u;
                            if (s >= t.left[y] && s < t.left[y] + t.count[y]) {
                                mask |= 1ull << (4u * u + (y - y0));
                                std::memcpy(&r[8u + 4u * u + (y - y0)], &t.w[(size_t)y * t.stride + (s - t.left[y])], sizeof(float));
                            }
                        }
                r[0] = s0;
                r[1] = (uint32_t)mask;
                r[2] = (uint32_t)(mask >> 32);
                r[3] = hi - 1u;
                r[4] = used;
                r[5] = (hi - lo + 7u) / 8u;  // ... in half chunks of 8 rows
            }
        }
    }
    const uint32_t hs = (t.stride + 3u) / 4u * 4u;
    if (hs / 4u > 8u) return;
    t.d2_hstride = hs;
    t.d2_hw.assign((size_t)out_n * hs, 0.0f);
    for (uint32_t x = 0; x < out_n; ++x)
        std::copy(t.w.begin() + (size_t)x * t.stride, t.w.begin() + (size_t)x * t.stride + t.count[x], t.d2_hw.begin() + (size_t)x * hs);
    // the widest strip whose source window is at most 64 quads (one per lane of the vertical pass), then evened out over the
    // strips it takes
    const uint32_t cap = 64u * down2_cols_per_lane(hs / 4u);
    for (uint32_t tw = std::min(cap, out_n); tw >= 1; --tw)
        if (tile_groups(t, out_n, tw) <= 64u) {
            const uint32_t strips = (out_n + tw - 1) / tw, even = (out_n + strips - 1) / strips;
            t.d2_tile_w = tile_groups(t, out_n, even) <= 64u ? even : tw;
            break;
        }
    for (uint32_t x0 = 0; t.d2_tile_w && x0 < out_n; x0 += t.d2_tile_w) {
        const uint32_t x1 = std::min(out_n, x0 + t.d2_tile_w), c0 = t.left[x0] & ~3u;
        t.d2_strips.push_back(c0);
        t.d2_strips.push_back((t.left[x1 - 1] + t.count[x1 - 1] - c0 + 3u) / 4u);
    }
}

int build_taps_host(uint32_t in_n, uint32_t out_n, int filter, TapsHost &t)
{
    float (*kern)(float) = nullptr;
    float support = 0.0f;
    switch (filter) {
    case KC_FILTER_NEAREST: kern = k_box; support = 0.0f; break;
    case KC_FILTER_TRIANGLE: kern = k_triangle; support = 1.0f; break;
    case KC_FILTER_CATMULLROM: kern = k_catmullrom; support = 2.0f; break;
    case KC_FILTER_GAUSSIAN: kern = k_gaussian; support = 3.0f; break;
    case KC_FILTER_LANCZOS3: kern = k_lanczos3; support = 3.0f; break;
    default: set_error("invalid ResizeFilter"); return KC_ERR_INVALID_ARG;
    }
    if (in_n == 0 || out_n == 0) {
        set_error("resize with zero extent");
        return KC_ERR_INVALID_ARG;
    }
    const float ratio = (float)in_n / (float)out_n;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = support * sratio;
    t.left.resize(out_n);
    t.count.resize(out_n);
    t.stride = 1;
    for (uint32_t o = 0; o < out_n; ++o) {
        const float input = ((float)o + 0.5f) * ratio;
        int64_t l = (int64_t)std::floor(input - src_support);
        if (l < 0) l = 0;
        if (l > (int64_t)in_n - 1) l = (int64_t)in_n - 1;
        int64_t r = (int64_t)std::ceil(input + src_support);
        if (r < l + 1) r = l + 1;
        if (r > (int64_t)in_n) r = (int64_t)in_n;
        t.left[o] = (uint32_t)l;
        t.count[o] = (uint32_t)(r - l);
        if (t.count[o] > t.stride) t.stride = t.count[o];
        if (o == 0 || t.count[o] < t.min_count) t.min_count = t.count[o];
    }
    t.w.assign((size_t)out_n * t.stride, 0.0f);
    for (uint32_t o = 0; o < out_n; ++o) {
        const float input = ((float)o + 0.5f) * ratio - 0.5f;
        float *w = &t.w[(size_t)o * t.stride];
        float sum = 0.0f;
        for (uint32_t j = 0; j < t.count[o]; ++j) {
            w[j] = kern(((float)(t.left[o] + j) - input) / sratio);
            sum += w[j];
        }
        for (uint32_t j = 0; j < t.count[o]; ++j) w[j] /= sum;
    }
    // The longest run of output indices whose windows are equally long, a whole number `step` of source samples apart, and
    // weighted bit-identically -- what integer-ratio down-sampling gives away from the border (resize_poly_kernel).
    t.reg_a = t.reg_b = t.reg_ages = t.reg_ratio = 0;
    if (out_n >= 2 && t.stride > KC_RESIZE_REG_TAPS) {
        uint32_t best_a = 0, best_b = 0, a0 = 0;
        auto same = [&](uint32_t x, uint32_t y) {
            return t.count[x] == t.count[y] && std::memcmp(&t.w[(size_t)x * t.stride], &t.w[(size_t)y * t.stride], t.count[x] * sizeof(float)) == 0;
        };
        for (uint32_t o = 1; o <= out_n; ++o) {
            const bool cont = o < out_n && same(o, a0) && (o == a0 + 1 || t.left[o] - t.left[o - 1] == t.left[a0 + 1] - t.left[a0]) &&
                              t.left[o] > t.left[o - 1];
            if (!cont) {
                if (o - a0 > best_b - best_a) {
                    best_a = a0;
                    best_b = o;
                }
                a0 = o;
            }
        }
        if (best_b - best_a >= 2) {
            const uint32_t step = t.left[best_a + 1] - t.left[best_a], n = t.count[best_a];
            if (step >= 2 && n % step == 0) {
                t.reg_a = best_a;
                t.reg_b = best_b;
                t.reg_ratio = step;
                t.reg_ages = n / step;
            }
        }
    }
    up_axis_build(in_n, out_n, t);
    t.d2_want = in_n > out_n && t.stride >= 4;  // a down-sampling axis with windows of 4 taps or more (Triangle from ratio 1.5 on)
    down2_build(out_n, t);
    return KC_OK;
}

// Copies e.host into one device block and points e.dev at it.  `pooled`: the block comes from the stream-ordered plane pool
// (band tables, which are evicted while kernels that read them may still be queued: giving the block back to the pool is
// safe in stream order and costs no device synchronisation, unlike hipFree); the pool's "in use" figure counts planes only.
static int taps_upload(TapsEntry &e, bool pooled = false)
{
    Context &c = ctx();
    const size_t nl = e.host.left.size() * sizeof(uint32_t);
    const size_t nw = e.host.w.size() * sizeof(float);
    const size_t nl_pad = (nl + 255) / 256 * 256;
    const size_t nw_pad = (nw + 32 + 255) / 256 * 256;  // + 32: register-tap loads past the last row
    const size_t nu = e.host.up_ok ? e.host.up_rows.size() * sizeof(float) : 0;
    const size_t nu_pad = (nu + 255) / 256 * 256;
    const size_t nuq = e.host.up_ok ? e.host.up_qrows.size() * sizeof(float) : 0;
    const size_t nuq_pad = (nuq + 255) / 256 * 256;
    const size_t nd2v = e.host.d2_vrec.size() * sizeof(uint32_t), nd2v_pad = (nd2v + 255) / 256 * 256;
    const size_t nd2h = e.host.d2_hw.size() * sizeof(float), nd2h_pad = (nd2h + 255) / 256 * 256;
    const size_t nd2s = e.host.d2_strips.size() * sizeof(uint32_t);
    e.dev_bytes = 2 * nl_pad + nw_pad + nu_pad + nuq_pad + nd2v_pad + nd2h_pad + (nd2s + 255) / 256 * 256;
    if (pooled) {
        KC_TRY(pool_alloc(e.dev_bytes, &e.dev_block));
        c.bytes_in_use -= e.dev_bytes;
    } else {
        KC_HIP(hipMalloc(&e.dev_block, e.dev_bytes));
    }
    char *base = (char *)e.dev_block;
    hipError_t err = hipMemcpyAsync(base, e.host.left.data(), nl, hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess) err = hipMemcpyAsync(base + nl_pad, e.host.count.data(), nl, hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess) err = hipMemcpyAsync(base + 2 * nl_pad, e.host.w.data(), nw, hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess && nu) err = hipMemcpyAsync(base + 2 * nl_pad + nw_pad, e.host.up_rows.data(), nu, hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess && nuq)
        err = hipMemcpyAsync(base + 2 * nl_pad + nw_pad + nu_pad, e.host.up_qrows.data(), nuq, hipMemcpyHostToDevice, c.stream);
    char *d2v = base + 2 * nl_pad + nw_pad + nu_pad + nuq_pad, *d2h = d2v + nd2v_pad, *d2s = d2h + nd2h_pad;
    if (err == hipSuccess && nd2s) err = hipMemcpyAsync(d2s, e.host.d2_strips.data(), nd2s, hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess && nd2v) err = hipMemcpyAsync(d2v, e.host.d2_vrec.data(), nd2v, hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess && nd2h) err = hipMemcpyAsync(d2h, e.host.d2_hw.data(), nd2h, hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess) err = hipStreamSynchronize(c.stream);
    if (err != hipSuccess) {
        if (pooled) {
            c.bytes_in_use += e.dev_bytes;
            pool_free(e.dev_block, e.dev_bytes);
        } else {
            (void)hipFree(e.dev_block);
        }
        e.dev_block = nullptr;
        return hip_fail(err, "upload tap table");
    }
    e.dev.left = (const uint32_t *)base;
    e.dev.count = (const uint32_t *)(base + nl_pad);
    e.dev.w = (const float *)(base + 2 * nl_pad);
    e.dev.stride = e.host.stride;
    e.host.d2_vrec_dev = nd2v ? (const uint32_t *)d2v : nullptr;
    e.host.d2_hw_dev = nd2h ? (const float *)d2h : nullptr;
    e.host.d2_strips_dev = nd2s ? (const uint32_t *)d2s : nullptr;
    if (nu) e.host.up.cls = (const float *)(base + 2 * nl_pad + nw_pad);
    if (nuq) e.host.up.qcls = (const float *)(base + 2 * nl_pad + nw_pad + nu_pad);
    return KC_OK;
}

static int get_taps(uint32_t in_n, uint32_t out_n, int filter, TapsEntry **out)
{
    Context &c = ctx();
    auto key = std::make_tuple(in_n, out_n, filter);
    auto it = c.taps.find(key);
    if (it != c.taps.end()) {
        *out = &it->second;
        return KC_OK;
    }
    TapsEntry e;
    KC_TRY(build_taps_host(in_n, out_n, filter, e.host));
    KC_TRY(taps_upload(e));
    auto ins = c.taps.emplace(key, std::move(e));
    *out = &ins.first->second;
    return KC_OK;
}

// Vertical tap table of a ROW BAND: output rows a .. b-1 of the logical out_n-row image (negative rows wrap
// around: a toroidal consumer such as HeightToNormal asks for row -1 = out_n - 1), read from a source band that
// holds rows src_y0 .. src_y0 + src_rows - 1 of the logical in_n-row source.  The weights are the full table's
// (same f32 values, same order): a band is computed exactly as the same rows of the whole image would be.
static int get_band_taps(uint32_t in_n, uint32_t out_n, int filter, int32_t a, int32_t b, int32_t src_y0, uint32_t src_rows,
                         TapsEntry **out)
{
    Context &c = ctx();
    auto key = std::make_tuple(in_n, out_n, filter, a, b, src_y0);
    auto it = c.band_taps.find(key);
    if (it != c.band_taps.end()) {
        // the key does not hold the number of rows the source band has: a table built for a taller band must not be
        // used on a shorter one without the check below (a kernel reading past the band is a GPU memory fault)
        uint32_t need_rows = 0;
        for (size_t i = 0; i < it->second.host.left.size(); ++i)
            need_rows = std::max(need_rows, it->second.host.left[i] + it->second.host.count[i]);
        if (need_rows > src_rows) {
            set_error("resize band: the source band does not hold the rows this output band needs (halo rows missing)");
            return KC_ERR_INVALID_ARG;
        }
        it->second.last_use = ++c.band_taps_clock;
        *out = &it->second;
        return KC_OK;
    }
    TapsEntry *full = nullptr;
    KC_TRY(get_taps(in_n, out_n, filter, &full));
    TapsEntry e;
    const uint32_t rows = (uint32_t)(b - a), stride = full->host.stride;
    e.host.stride = stride;
    e.host.left.resize(rows);
    e.host.count.resize(rows);
    e.host.w.assign((size_t)rows * stride, 0.0f);
    e.host.min_count = stride;
    for (uint32_t i = 0; i < rows; ++i) {
        const int64_t logical = (int64_t)a + i;
        const uint32_t oy = (uint32_t)(((logical % (int64_t)out_n) + out_n) % out_n);
        const int64_t rel = (int64_t)full->host.left[oy] - src_y0;
        const uint32_t n = full->host.count[oy];
        if (rel < 0 || rel + n > src_rows) {
            set_error("resize band: the source band does not hold the rows this output band needs (halo rows missing)");
            return KC_ERR_INVALID_ARG;
        }
        e.host.left[i] = (uint32_t)rel;
        e.host.count[i] = n;
        if (n < e.host.min_count) e.host.min_count = n;
        std::copy(full->host.w.begin() + (size_t)oy * stride, full->host.w.begin() + (size_t)(oy + 1) * stride,
                  e.host.w.begin() + (size_t)i * stride);
    }
    e.host.d2_want = full->host.d2_want;
    down2_build(rows, e.host);
    KC_TRY(taps_upload(e, true));
    // Bands come in a handful of shapes per graph; bound what a long-lived process keeps.  The least recently used table
    // goes, ONE at a time, and its block returns to the stream-ordered pool: kernels already enqueued may still read it (whoever
    // gets the block next is enqueued behind them), and a synchronous hipFree in the middle of an evaluation would stall the
    // device (nobody holds a TapsEntry across a call that can get here: every caller launches right after its look-up,
    // under the context lock).
    e.last_use = ++c.band_taps_clock;
    if (c.band_taps.size() >= 64) {
        auto victim = c.band_taps.begin();
        for (auto it = c.band_taps.begin(); it != c.band_taps.end(); ++it)
            if (it->second.last_use < victim->second.last_use) victim = it;
        c.bytes_in_use += victim->second.dev_bytes;
        pool_free(victim->second.dev_block, victim->second.dev_bytes);
        c.band_taps.erase(victim);
    }
    auto ins = c.band_taps.emplace(key, std::move(e));
    *out = &ins.first->second;
    return KC_OK;
}

// LDS pitch in floats for such tiles; an odd group count staggers consecutive tile rows over the banks.
static uint32_t tile_pitch(const TapsHost &h, uint32_t out_n, uint32_t tile_w) { return 4u * (tile_groups(h, out_n, tile_w) | 1u); }

ResizeMemoScope::ResizeMemoScope()
{
    Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    c.memo_depth++;
}

ResizeMemoScope::~ResizeMemoScope()
{
    Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    if (--c.memo_depth == 0) {
        for (auto &kv : c.resize_memo) {
            plane_release(std::get<0>(kv.first));
            plane_release(kv.second);
        }
        c.resize_memo.clear();
    }
}

static int resize_plane_uncached(kc_plane *src, kc_size size, int filter, kc_plane **out);

static int resize_plane(kc_plane *src, kc_size size, int filter, kc_plane **out)
{
    Context &c = ctx();
    if (c.memo_depth == 0) return resize_plane_uncached(src, size, filter, out);
    auto key = std::make_tuple(src, size.width, size.height, filter);
    auto it = c.resize_memo.find(key);
    if (it != c.resize_memo.end()) {
        plane_retain(it->second);
        *out = it->second;
        return KC_OK;
    }
    KC_TRY(resize_plane_uncached(src, size, filter, out));
    if (c.resize_memo.size() >= 32) {  // bound the HBM a long evaluation can pin
        for (auto &kv : c.resize_memo) {
            plane_release(std::get<0>(kv.first));
            plane_release(kv.second);
        }
        c.resize_memo.clear();
    }
    plane_retain(src);
    plane_retain(*out);
    c.resize_memo.emplace(key, *out);
    return KC_OK;
}

// Picks the output tile for one resample: the first candidate whose vertical-pass intermediate
// (tile_h rows of the tile's source-column window) fits the LDS budget.  Wide tiles make every tile
// row one long run of 16-byte stores; the budget keeps several workgroups resident per CU.
struct TileChoice {
    uint32_t tile_w = 0, tile_h = 0, ncp = 0;
    bool ok = false;
    bool down = false;  // resize_down_kernel (both axes wider than the register-tap forms)
    bool poly = false;  // ... and its vertical table is regular: resize_poly_kernel
};

static bool tile_fits(const TapsEntry &tv, const TapsEntry &th, kc_size size, uint32_t tw, uint32_t tht, size_t budget, TileChoice &t)
{
    if (tw % 4 != 0 || tw > 1024 || tw == 0 || tht == 0 || tht > 64 || 256u % (tw / 4) != 0) return false;
    const uint32_t ncp = tile_pitch(th.host, size.width, tw);
    if (resize_lds_bytes(tht, ncp, tv.dev.stride, tw, th.dev.stride) > budget) return false;
    t.tile_w = tw;
    t.tile_h = tht;
    t.ncp = ncp;
    t.ok = true;
    return true;
}

static TileChoice choose_tile(const TapsEntry &tv, const TapsEntry &th, kc_size size)
{
    Context &c = ctx();
    TileChoice t;
    // Both axes down-sampled: the wave-uniform form.  Its vertical pass deals 64 column quads to a wave, so the widest
    // tile whose source window is at most 64 quads wastes no lanes; a wave walks 4 (tile of 16) or 8 (tile of 32) rows.
    if (th.dev.stride > KC_RESIZE_REG_TAPS && tv.dev.stride > KC_RESIZE_REG_TAPS && c.resize_mode != 2) {
        const uint32_t rows = c.resize_tile_h == 32 ? 32u : 16u;  // KC_RESIZE_TILE_H=32: tuning
        for (uint32_t tw = 64; tw >= 4; tw -= 4) {
            const uint32_t groups = tile_groups(th.host, size.width, tw);
            if (groups > 64) continue;  // the intermediate rows hold 256 floats
            if (resize_down_lds_bytes(rows, 4u * groups, tw, th.dev.stride) > 64 * 1024) continue;
            t.tile_w = tw;
            t.tile_h = rows;
            t.ncp = 4u * groups;
            t.ok = t.down = true;
            const TapsHost &hv = tv.host;
            t.poly = rows == 16 && c.resize_mode != 1 && (hv.reg_ages == 2 || hv.reg_ages == 4 || hv.reg_ages == 6) &&
                     (hv.reg_ratio == 2 || hv.reg_ratio == 4 || hv.reg_ratio == 8) && hv.reg_b - hv.reg_a >= 16 && hv.reg_a <= 16 &&
                     size.height - (hv.reg_a + (hv.reg_b - hv.reg_a) / 4 * 4) <= 48;
            return t;
        }
    }
    if (c.resize_tile_w > 0 && c.resize_tile_h > 0 &&  // tuning override (KC_RESIZE_TILE_W / _H)
        tile_fits(tv, th, size, (uint32_t)c.resize_tile_w, (uint32_t)c.resize_tile_h, 64 * 1024, t))
        return t;
    // Wide horizontal windows (down-sampling): the vertical pass re-reads window rows per output row, so small tiles --
    // many workgroups, short dependent chains -- win (profiles/resize_tile_sweep.py).  With the intermediate rows
    // swizzled (no LDS bank conflicts in the horizontal pass) 64 x 8 is the best all-rounder: Lanczos3 4x 37.9 -> 36.3 us,
    // CatmullRom 3x 45.8 -> 33.3, Gaussian 4.3x 25.9 -> 25.2, Triangle 8x 19.9 -> 20.6 (profiles/r02_wide_tiles.txt).
    static const uint32_t wide[][2] = { { 64, 8 }, { 32, 4 }, { 16, 4 }, { 8, 4 }, { 4, 4 } };
    static const uint32_t tiles[][2] = { { 1024, 16 }, { 1024, 8 }, { 512, 16 }, { 512, 8 }, { 256, 16 }, { 256, 8 },
                                         { 128, 16 },  { 128, 8 },  { 64, 8 },   { 32, 8 },  { 16, 8 },   { 16, 4 },
                                         { 8, 4 },     { 4, 4 } };
    if (th.dev.stride > KC_RESIZE_REG_TAPS)
        for (auto &tl : wide)
            if (tile_fits(tv, th, size, tl[0], tl[1], 64 * 1024, t)) return t;
    for (size_t budget : { (size_t)40 * 1024, (size_t)64 * 1024 })
        for (auto &tl : tiles)
            if (tile_fits(tv, th, size, tl[0], tl[1], budget, t)) return t;
    return t;
}

// Integer-ratio up-sampling on both axes (upsample.h): tile and LDS pitch for upsample_chain_tile, or false.
static bool up_plan(const TapsEntry &tv, const TapsEntry &th, UpsampleArgs &u)
{
    Context &c = ctx();
    if (c.resize_mode >= 3 || !tv.host.up_ok || !th.host.up_ok) return false;
    u.H = th.host.up;
    u.V = tv.host.up;
    if (u.H.taps != u.V.taps || !u.H.qcls || !u.V.cls) return false;
    if (u.H.ratio % 4 != 0 && !(u.H.ratio == 2 && u.H.n_out % 4 == 0)) return false;  // (ratio 2: "half quads", upsample.h)
    const uint32_t dw = u.H.n_out, R = u.H.ratio, T = u.H.taps;
    // widest tile that wastes the fewest threads on columns past the image (narrower tiles are taller: a thread always
    // owns 4 columns x KC_UPSAMPLE_ROWS rows)
    uint32_t best = 0, best_pad = 0;
    for (uint32_t tw : { 1024u, 512u, 256u, 128u, 64u, 32u, 16u }) {
        const uint32_t pad = (dw + tw - 1) / tw * tw;
        if (!best || pad < best_pad) {
            best = tw;
            best_pad = pad;
        }
    }
    if (c.resize_tile_w > 0 && c.resize_tile_w % 4 == 0 && c.resize_tile_w <= 1024 && 256 % (c.resize_tile_w / 4) == 0)
        best = (uint32_t)c.resize_tile_w;  // KC_RESIZE_TILE_W: tuning
    u.tile_w = best;
    uint32_t quads = 1;  // the widest window any tile needs, in source quads, exactly as the kernel lays it out
    for (uint32_t x0 = 0; x0 < dw; x0 += u.tile_w) {
        const uint32_t x1 = std::min(dw, x0 + u.tile_w);
        const int cq0 = ((int)(x0 / R) - u.H.off) >> 2;
        const int last = (int)((x1 - 1) / R) - u.H.off + (int)(T - 1);
        quads = std::max(quads, (uint32_t)((last >> 2) - cq0) + 1u);
    }
    u.ncp = 4u * (quads | 1u);  // an odd quad count staggers consecutive rows over the LDS banks
    if (quads > 256 || (std::max(R >> 2, 1u) + u.H.qb_lo + u.H.qb_hi) * T > 256) return false;  // one vertical item / one class quad per thread
    u.chunk = 1;  // the most rows that share a window and divide the tile's rows (a multiple of KC_UPSAMPLE_ROWS)
    for (uint32_t d : { 8u, 4u, 2u })
        if (u.V.ratio % d == 0 && KC_UPSAMPLE_ROWS % d == 0) {
            u.chunk = d;
            break;
        }
    return upsample_lds_bytes(u) <= 64 * 1024;
}

// Runs the resample srcs[i] -> dsts[i] (all resident; equal source sizes, equal target sizes) with the given tap
// tables: one launch for the whole batch in the tiled form, two per plane in the two-pass form.
static int resize_run_taps(kc_plane *const *srcs, kc_plane *const *dsts, int n, TapsEntry *tv, TapsEntry *th)
{
    Context &c = ctx();
    const kc_plane *s0 = srcs[0];
    const kc_size size{ dsts[0]->w, dsts[0]->h };
    // Tiled single pass when a tile's vertical-pass intermediate and tap tables fit in LDS; very wide
    // windows fall back to two passes through an HBM intermediate (KC_RESIZE_MODE=3 forces them).
    UpsampleArgs ua{};
    if (up_plan(*tv, *th, ua)) {
        UpsamplePlanes up{};
        for (int i = 0; i < n; ++i) {
            up.samp_src[i] = srcs[i]->dptr;
            up.samp_pitch[i] = (uint32_t)(srcs[i]->pitch / 4);
            up.out[i] = dsts[i]->dptr;
            up.out_pitch[i] = (uint32_t)(dsts[i]->pitch / 16);
        }
        up.nt_mask = cache_policy_mask(0, (uint64_t)n * 4 * size.width * size.height, 0);  // the small source stays cacheable
        hipError_t e = launch_upsample(up, n, ua, c.stream);
        if (e != hipSuccess) return hip_fail(e, "launch_upsample");
        c.launches++;
        c.counters["upsample_launches"]++;
        c.alg_bytes += (uint64_t)n * 4 * ((uint64_t)s0->w * s0->h + (uint64_t)size.width * size.height);
        return KC_OK;
    }
    if (c.resize_mode != 3) {
        const TileChoice t = choose_tile(*tv, *th, size);
        if (t.ok) {
            ResizePlanes rp{};
            for (int i = 0; i < n; ++i) {
                rp.src[i] = srcs[i]->dptr;
                rp.dst[i] = dsts[i]->dptr;
                rp.spitch[i] = (uint32_t)(srcs[i]->pitch / 4);
                rp.dpitch[i] = (uint32_t)(dsts[i]->pitch / 4);
            }
            // both axes down-sampled: the wave-private form where its tables exist (down2.hip)
            // ... except where the streaming kernel of integer ratios is the faster one: ratios 4 and 8 (Lanczos3 4096^2 -> 1024^2
            // 23.5 against 26.7 us, CatmullRom 18.1 / 21.4; at ratio 2 down2 wins, 26.3 / 30.4 -- profiles/r03_down2_ab.txt)
            // tiles in XCD order while source and result stay in the Infinity Cache (the budget of the cache policy)
            const bool fits_cache = (uint64_t)n * 4 * ((uint64_t)s0->w * s0->h + (uint64_t)size.width * size.height) <= (208ull << 20);
            // integer ratios: two waves to a band's strip (resize_poly2_kernel)
            // (where it measures faster than the forms below -- profiles/r04_poly2_sweep.txt: ratio 8 with windows of 4 or 6 ages,
            // Gaussian 4096^2 -> 512^2 27.6 -> 25.3 us, 8192^2 -> 1024^2 82.5 -> 72.5; at ratio 4 and 2 it is behind resize_poly_kernel
            // and resize_down2_kernel, 24.8 against 21.2 us and 34.3 against 24.5; kc_set_option("poly2_min_ratio") moves the line)
            if (t.poly && c.poly2 && tv->host.reg_ratio >= (uint32_t)c.poly2_min_ratio && tv->host.reg_ages >= 4 && th->host.p2_tile_w) {
                hipError_t e2 = launch_resize_poly2(rp, n, size.width, size.height, tv->dev, th->dev, th->host.p2_tile_w, t.tile_w, t.ncp,
                                                    tv->host.reg_a, tv->host.reg_b, tv->host.reg_ages, tv->host.reg_ratio, fits_cache, c.stream);
                if (e2 != hipSuccess) return hip_fail(e2, "launch_resize_poly2");
                c.launches++;
                c.counters["poly2_launches"]++;
                c.alg_bytes += (uint64_t)n * 4 * ((uint64_t)s0->w * s0->h + (uint64_t)size.width * size.height);
                return KC_OK;
            }
            const bool poly_first = t.poly && tv->host.reg_ratio >= 4;
            if (c.down2 > (poly_first ? 1 : 0) && tv->host.d2_nc && tv->host.d2_vrec_dev && th->host.d2_tile_w &&
                th->host.d2_hw_dev && th->host.d2_strips_dev) {
                Down2Args a{};
                a.vrec = tv->host.d2_vrec_dev;
                a.nc = tv->host.d2_nc;
                a.hleft = th->dev.left;
                a.hcount = th->dev.count;
                a.hw = th->host.d2_hw_dev;
                a.hstride = th->host.d2_hstride;
                a.strips = th->host.d2_strips_dev;
                a.tile_w = th->host.d2_tile_w;
                a.dw = size.width;
                a.dh = size.height;
                a.xcd_per = fits_cache ? 1u : 0u;
                a.by_rows = c.down2_by_rows < 0 ? (tv->host.d2_nc > 1 ? 1u : 0u) : (c.down2_by_rows ? 1u : 0u);
                hipError_t e2 = launch_resize_down2(rp, n, a, c.stream);
                if (e2 != hipSuccess) return hip_fail(e2, "launch_resize_down2");
                c.launches++;
                c.counters["down2_launches"]++;
                c.alg_bytes += (uint64_t)n * 4 * ((uint64_t)s0->w * s0->h + (uint64_t)size.width * size.height);
                return KC_OK;
            }
            hipError_t e = t.poly ? launch_resize_poly(rp, n, size.width, size.height, tv->dev, th->dev, t.tile_w, t.ncp, tv->host.reg_a,
                                                       tv->host.reg_b, tv->host.reg_ages, tv->host.reg_ratio, c.stream)
                           : t.down ? launch_resize_down(rp, n, size.width, size.height, tv->dev, th->dev, t.tile_w, t.tile_h, t.ncp, c.stream)
                                  : launch_resize_lds(rp, n, size.width, size.height, tv->dev, th->dev, th->host.min_count, t.tile_w,
                                                      t.tile_h, t.ncp, c.stream);
            if (e != hipSuccess) return hip_fail(e, "launch_resize_lds");
            c.launches++;
            c.alg_bytes += (uint64_t)n * 4 * ((uint64_t)s0->w * s0->h + (uint64_t)size.width * size.height);
            return KC_OK;
        }
    }
    for (int i = 0; i < n; ++i) {
        kc_plane *tmp = nullptr;
        KC_TRY(plane_new_mem(s0->w, size.height, &tmp));
        const uint32_t tpitch = (uint32_t)(tmp->pitch / 4);
        hipError_t e = launch_resize_vertical(srcs[i]->dptr, (uint32_t)(srcs[i]->pitch / 4), s0->w, tmp->dptr, tpitch,
                                              size.height, tv->dev, c.stream);
        if (e == hipSuccess)
            e = launch_resize_horizontal(tmp->dptr, tpitch, dsts[i]->dptr, (uint32_t)(dsts[i]->pitch / 4), size.width,
                                         size.height, th->dev, c.stream);
        plane_release(tmp);
        if (e != hipSuccess) return hip_fail(e, "launch_resize two-pass");
        c.launches += 2;
        c.alg_bytes += 4 * ((uint64_t)s0->w * s0->h + 2 * (uint64_t)s0->w * size.height + (uint64_t)size.width * size.height);
    }
    return KC_OK;
}

static int resize_run(kc_plane *const *srcs, kc_plane *const *dsts, int n, int filter)
{
    TapsEntry *tv = nullptr, *th = nullptr;
    KC_TRY(get_taps(srcs[0]->h, dsts[0]->h, filter, &tv));
    KC_TRY(get_taps(srcs[0]->w, dsts[0]->w, filter, &th));
    return resize_run_taps(srcs, dsts, n, tv, th);
}

// Row-band form of resize_image for resident planes (bands.cpp): srcs[i] holds rows src_y0 .. of a logical
// (srcs[i]->w x src_h_full) plane; outs[i] receives rows a .. b-1 (negative = wrapped) of its resample to dst_full.
int resize_planes_band(kc_plane *const *srcs, int n, int32_t src_y0, uint32_t src_h_full, kc_size dst_full, int32_t a, int32_t b,
                       int filter, kc_plane **outs)
{
    KC_TRY(need_init());
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    if (n < 1 || n > 4 || b <= a) {
        set_error("resize band: bad arguments");
        return KC_ERR_INVALID_ARG;
    }
    TapsEntry *tv = nullptr, *th = nullptr;
    KC_TRY(get_band_taps(src_h_full, dst_full.height, filter, a, b, src_y0, srcs[0]->h, &tv));
    KC_TRY(get_taps(srcs[0]->w, dst_full.width, filter, &th));
    int s = KC_OK;
    for (int i = 0; i < n; ++i) outs[i] = nullptr;
    for (int i = 0; i < n && s == KC_OK; ++i) s = plane_new_mem(dst_full.width, (uint32_t)(b - a), &outs[i]);
    if (s == KC_OK) s = resize_run_taps(srcs, outs, n, tv, th);
    if (s != KC_OK)
        for (int i = 0; i < n; ++i) {
            plane_release(outs[i]);
            outs[i] = nullptr;
        }
    return s;
}

// RESIZE -> MEM through the plain resize kernels.  Planes that resample equally sized sources to the
// same size with the same filter (the planes of an image) share launches, four at a time.
int resize_force_many(kc_plane *const *planes, int n)
{
    std::vector<kc_plane *> todo;
    for (int i = 0; i < n; ++i) {
        kc_plane *p = planes[i];
        if (!p || p->kind != kc_plane::RESIZE) continue;
        bool dup = false;
        for (auto *q : todo) dup |= (q == p);
        if (!dup) todo.push_back(p);
    }
    if (todo.empty()) return KC_OK;
    KC_TRY(need_init());
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    std::vector<bool> done(todo.size(), false);
    for (size_t i = 0; i < todo.size(); ++i) {
        if (done[i]) continue;
        kc_plane *group[4], *srcs[4], *dsts[4] = { nullptr, nullptr, nullptr, nullptr };
        int g = 0;
        for (size_t j = i; j < todo.size() && g < 4; ++j) {
            kc_plane *q = todo[j], *p = todo[i];
            if (done[j] || q->w != p->w || q->h != p->h || q->rz_filter != p->rz_filter || q->rz_src->w != p->rz_src->w ||
                q->rz_src->h != p->rz_src->h)
                continue;
            group[g] = q;
            srcs[g] = q->rz_src;
            done[j] = true;
            ++g;
        }
        int s = KC_OK;
        for (int k = 0; k < g && s == KC_OK; ++k) s = plane_new_mem(group[k]->w, group[k]->h, &dsts[k]);
        if (s == KC_OK) s = resize_run(srcs, dsts, g, group[0]->rz_filter);
        if (s != KC_OK) {
            for (int k = 0; k < g; ++k) plane_release(dsts[k]);
            return s;
        }
        for (int k = 0; k < g; ++k) {
            kc_plane *p = group[k], *dst = dsts[k];
            p->kind = kc_plane::MEM;
            p->dptr = dst->dptr;
            p->pitch = dst->pitch;
            p->bytes = dst->bytes;
            p->owned = true;
            dst->owned = false;
            dst->dptr = nullptr;
            plane_release(dst);
            plane_release(p->rz_src);
            p->rz_src = nullptr;
        }
    }
    return KC_OK;
}

int resize_force(kc_plane *p) { return resize_force_many(&p, 1); }

// Fused resample + chain (see kc_runtime.hpp).  Eligible: every channel's resampled operand uses
// the same tap tables (same source size and filter), at most 4 horizontal taps (held in
// registers), a {+,-,*} program of at most 16 steps, and an LDS tile exists.
int chain_resize_launch(const ChainProgram &P, int batch, int mode, kc_plane *const *sampled, bool *launched)
{
    Context &c = ctx();
    *launched = false;
    if (!c.fusion || c.resize_mode == 3 || mode != 0 || P.n_ops > 16 || P.n_in < 1 || P.n_in > 4) return KC_OK;
    const kc_plane *s0 = sampled[0];
    const kc_size size{ s0->w, s0->h };
    TapsEntry *tv = nullptr, *th = nullptr;
    KC_TRY(get_taps(s0->rz_src->h, size.height, s0->rz_filter, &tv));
    KC_TRY(get_taps(s0->rz_src->w, size.width, s0->rz_filter, &th));
    UpsampleArgs ua{};
    if (up_plan(*tv, *th, ua) && ua.H.taps <= 3) {
        // the chain as straight-line code if that kernel has been compiled (specialize.cpp), otherwise the interpreter
        bool spec = false;
        hipError_t e = launch_upsample_chain_specialized(P, batch, ua, c.stream, &spec);
        if (e == hipSuccess && !spec) e = launch_upsample_chain(P, batch, ua, c.stream);
        if (e != hipSuccess) return hip_fail(e, "launch_upsample_chain");
        c.counters["upsample_chain_launches"]++;
        *launched = true;
        return KC_OK;
    }
    if (th->host.stride > 4) return KC_OK;
    const TileChoice t = choose_tile(*tv, *th, size);
    if (!t.ok) return KC_OK;
    hipError_t e = launch_resize_chain(P, batch, size.width, size.height, tv->dev, th->dev, t.tile_w, t.tile_h, t.ncp, c.stream);
    if (e != hipSuccess) return hip_fail(e, "launch_resize_chain");
    c.counters["resize_chain_launches"]++;
    *launched = true;
    return KC_OK;
}

static int resize_plane_uncached(kc_plane *src, kc_size size, int filter, kc_plane **out)
{
    // A 1x1 source has a single tap whose normalised weight is w/w = 1, in both passes:
    // t = 0.0 + v*1.0 (vertical), u = 0.0 + t*1.0 then clamp (horizontal) -- a constant plane.
    if (src->w == 1 && src->h == 1 && src->kind == kc_plane::CONST) {
        float t = 0.0f;
        t += src->cval * 1.0f;
        float u = 0.0f;
        u += t * 1.0f;
        if (u < 0.0f) u = 0.0f;
        else if (u > 1.0f) u = 1.0f;
        *out = plane_new_const(size.width, size.height, u);
        return KC_OK;
    }
    if (filter < KC_FILTER_NEAREST || filter > KC_FILTER_LANCZOS3) {
        set_error("invalid ResizeFilter");
        return KC_ERR_INVALID_ARG;
    }
    KC_TRY(need_init());
    KC_TRY(plane_materialize(src));
    // Deferred: a Mix chain that consumes the result resamples inside its own kernel; any other
    // consumer runs the resize kernel on first use (with fusion switched off, resize_image does).
    kc_plane *p = new kc_plane();
    p->w = size.width;
    p->h = size.height;
    p->kind = kc_plane::RESIZE;
    p->rz_src = src;
    p->rz_filter = filter;
    plane_retain(src);
    *out = p;
    return KC_OK;
}

// One imageops::resize call per plane, as src/shared.rs:156-201 does (1 for Gray, 4 for Rgba --
// aliased planes, e.g. Gray->Rgba [p, p, p, ones], are resampled once and re-aliased).
int resize_image(kc_image *src, kc_size size, int filter, kc_image **out)
{
    KC_PROF("resize_image");
    if (size.width == 0 || size.height == 0) {
        set_error("resize to zero extent");
        return KC_ERR_INVALID_ARG;
    }
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    kc_plane *p[4] = { nullptr, nullptr, nullptr, nullptr };
    int s = KC_OK;
    for (int i = 0; i < src->n && s == KC_OK; ++i) {
        for (int j = 0; j < i; ++j)
            if (src->planes[j] == src->planes[i]) {
                p[i] = p[j];
                plane_retain(p[i]);
                break;
            }
        if (!p[i]) s = resize_plane(src->planes[i], size, filter, &p[i]);
    }
    if (s == KC_OK && !ctx().fusion) s = resize_force_many(p, src->n);  // one launch for the image's planes
    if (s == KC_OK) *out = image_new(src->n, p);
    for (int i = 0; i < src->n; ++i) plane_release(p[i]);
    return s;
}

}  // namespace kc
