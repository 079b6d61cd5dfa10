// Node operators: the bodies behind process_node_internal (src/node/node_type.rs:98-138) and the
// SlotImage conversions they use.  Host logic only decides WHICH planes feed which kernel; every
// pixel is produced by a kernel in kernels.hip.
#include <cstdlib>
#include <cstring>

#include "kc_runtime.hpp"

namespace kc {

// SlotImage::from_value, src/slot_image.rs:28-64 -- [v, v, v, 1.0] / [v] as constant planes.
int image_from_value(kc_size size, float v, bool rgba, kc_image **out)
{
    if (size.width == 0 || size.height == 0) {
        set_error("from_value with zero extent");
        return KC_ERR_INVALID_ARG;
    }
    kc_plane *p[4];
    int n = rgba ? 4 : 1;
    for (int i = 0; i < n; ++i) p[i] = plane_new_const(size.width, size.height, (rgba && i == 3) ? 1.0f : v);
    *out = image_new(n, p);
    for (int i = 0; i < n; ++i) plane_release(p[i]);
    return KC_OK;
}

// SlotImage::as_type, src/slot_image.rs:212-256.
int image_as_type(kc_image *img, bool rgba, kc_image **out)
{
    if (img->is_rgba() == rgba) {
        image_retain(img);
        *out = img;
        return KC_OK;
    }
    if (!img->is_rgba()) {
        // Gray -> Rgba: [p, p, p, ones] -- the three colour planes alias the input (Arc::clone)
        kc_plane *ones = plane_new_const(img->w(), img->h(), 1.0f);
        kc_plane *p[4] = { img->planes[0], img->planes[0], img->planes[0], ones };
        *out = image_new(4, p);
        plane_release(ones);
        return KC_OK;
    }
    // Rgba -> Gray: ((r + g) + b) / 3. -- expressed as a pointwise chain: Add g, Add b, Divide 3
    kc_plane *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;
    KC_TRY(plane_mix(KC_MIX_ADD, img->planes[0], img->planes[1], &t1));
    int s = plane_mix(KC_MIX_ADD, t1, img->planes[2], &t2);
    plane_release(t1);
    if (s != KC_OK) return s;
    kc_plane *three = plane_new_const(img->w(), img->h(), 3.0f);
    s = plane_mix(KC_MIX_DIVIDE, t2, three, &t3);
    plane_release(t2);
    plane_release(three);
    if (s != KC_OK) return s;
    *out = image_new(1, &t3);
    plane_release(t3);
    if (!ctx().fusion) KC_TRY(image_force(*out));  // unfused mode: as_type is its own pass, like the reference's loop
    return KC_OK;
}

// deconstruct_image + read_slot_image, src/shared.rs:16-56,218-261.
int image_from_u8(const uint8_t *host, uint32_t w, uint32_t h, int channels, kc_image **out)
{
    KC_TRY(need_init());
    if (!host || w == 0 || h == 0 || channels < 1 || channels > 4) {
        set_error("image_from_u8: bad arguments");
        return KC_ERR_INVALID_ARG;
    }
    Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    const size_t nbytes = (size_t)w * h * channels;
    const size_t block = (nbytes + 255) / 256 * 256;
    void *staging = nullptr;
    KC_TRY(pool_alloc(block, &staging));
    kc_plane *p[4] = { nullptr, nullptr, nullptr, nullptr };
    float *dp[4] = { nullptr, nullptr, nullptr, nullptr };
    int s = KC_OK;
    for (int i = 0; i < 4 && s == KC_OK; ++i) {
        if (i < channels) {
            s = plane_new_mem(w, h, &p[i]);
            if (s == KC_OK) dp[i] = p[i]->dptr;
        } else {
            p[i] = plane_new_const(w, h, i == 3 ? 1.0f : 0.0f);
        }
    }
    if (s == KC_OK) {
        hipError_t e = hipSuccess;
        // The caller may free `host` as soon as we return.  Small images (the reference's own sizes) go through a pinned
        // ring slot and the call does not wait for the stream; larger ones are copied from `host` directly and waited for.
        Context::UploadSlot *slot = nullptr;
        static const bool ring_on = !std::getenv("KC_UPLOAD_RING") || std::atoi(std::getenv("KC_UPLOAD_RING")) != 0;  // 0: A/B
        if (ring_on && nbytes <= Context::kUploadSlotMax) {
            slot = &c.upload_ring[c.upload_next];
            c.upload_next = (c.upload_next + 1) % Context::kUploadSlots;
            if (slot->bytes < nbytes) {
                if (slot->host) {
                    if (slot->copied) (void)hipEventSynchronize(slot->copied);
                    (void)hipHostFree(slot->host);
                    slot->host = nullptr;
                    slot->bytes = 0;
                }
                const size_t want = std::max(nbytes, (size_t)256 << 10);
                e = hipHostMalloc(&slot->host, want, hipHostMallocDefault);
                if (e == hipSuccess) slot->bytes = want;
                else slot->host = nullptr;
            }
            if (e == hipSuccess && !slot->copied) e = hipEventCreateWithFlags(&slot->copied, hipEventDisableTiming);
            else if (e == hipSuccess) e = hipEventSynchronize(slot->copied);  // the slot's previous upload
            if (e == hipSuccess) {
                std::memcpy(slot->host, host, nbytes);
                e = hipMemcpyAsync(staging, slot->host, nbytes, hipMemcpyHostToDevice, c.stream);
                if (e == hipSuccess) e = hipEventRecord(slot->copied, c.stream);
            }
        } else {
            e = hipMemcpyAsync(staging, host, nbytes, hipMemcpyHostToDevice, c.stream);
        }
        if (e == hipSuccess)
            e = launch_from_u8((const uint8_t *)staging, channels, w, h, dp, (uint32_t)(p[0]->pitch / 4),
                               cache_policy_mask((uint64_t)w * h * channels, (uint64_t)w * h * 4 * channels, 1), c.stream);
        if (e == hipSuccess && !slot) e = hipStreamSynchronize(c.stream);
        if (e != hipSuccess) s = hip_fail(e, "image_from_u8");
        else {
            c.launches++;
            c.alg_bytes += (uint64_t)w * h * channels * 5;  // 1 B read + 4 B written per sample
        }
    }
    pool_free(staging, block);
    if (s == KC_OK) *out = image_new(4, p);
    for (int i = 0; i < 4; ++i) plane_release(p[i]);
    return s;
}

// SlotImage::to_u8 / to_u8_srgb, src/slot_image.rs:141-207.
int image_to_u8(kc_image *img, bool srgb, uint8_t *host)
{
    KC_TRY(need_init());
    Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    KC_TRY(image_force(img));
    const uint32_t w = img->w(), h = img->h();
    const size_t nbytes = (size_t)w * h * 4;
    const size_t block = (nbytes + 255) / 256 * 256;
    void *staging = nullptr;
    KC_TRY(pool_alloc(block, &staging));
    Operand o[4];
    for (int i = 0; i < 4; ++i) o[i] = plane_operand(img->planes[img->is_rgba() ? i : 0]);
    uint32_t n_res = 0;
    for (int i = 0; i < (img->is_rgba() ? 4 : 1); ++i) n_res += o[i].ptr != nullptr;
    hipError_t e = launch_to_u8(o[0], o[1], o[2], o[3], img->is_rgba() ? 0 : 1, srgb ? 1 : 0, w, h, (uint8_t *)staging,
                                cache_policy_mask((uint64_t)w * h * 4 * n_res, nbytes, n_res ? n_res : 1), c.stream);
    if (e == hipSuccess) {
        c.launches++;
        int resident = 0;
        for (int i = 0; i < (img->is_rgba() ? 4 : 1); ++i) resident += o[i].ptr != nullptr;
        c.alg_bytes += (uint64_t)w * h * 4 * (resident + 1);
        e = hipMemcpyAsync(host, staging, nbytes, hipMemcpyDeviceToHost, c.stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    pool_free(staging, block);
    if (e != hipSuccess) return hip_fail(e, "image_to_u8");
    return KC_OK;
}

// calculate_size, src/shared.rs:61-139.
int calculate_size(int policy, const kc_size *sizes, int n, int slot_index, kc_size specific, kc_size *out)
{
    auto px = [](kc_size s) { return (uint32_t)(s.width * s.height); };  // Size::pixel_count multiplies in u32
    switch (policy) {
    case KC_POLICY_MOST_PIXELS: {
        if (n == 0) {
            *out = kc_size{ 1, 1 };
            return KC_OK;
        }
        int best = 0;  // Iterator::max_by returns the last maximum
        for (int i = 1; i < n; ++i)
            if (px(sizes[i]) >= px(sizes[best])) best = i;
        *out = sizes[best];
        return KC_OK;
    }
    case KC_POLICY_LEAST_PIXELS: {
        if (n == 0) {
            set_error("LeastPixels with no inputs (the reference unwraps None here)");
            return KC_ERR_NODE_PROCESSING;
        }
        int best = 0;  // Iterator::min_by returns the first minimum
        for (int i = 1; i < n; ++i)
            if (px(sizes[i]) < px(sizes[best])) best = i;
        *out = sizes[best];
        return KC_OK;
    }
    case KC_POLICY_LARGEST_AXES: {
        kc_size s{ 0, 0 };
        for (int i = 0; i < n; ++i) {
            if (sizes[i].width > s.width) s.width = sizes[i].width;
            if (sizes[i].height > s.height) s.height = sizes[i].height;
        }
        *out = s;
        return KC_OK;
    }
    case KC_POLICY_SMALLEST_AXES: {
        kc_size s{ UINT32_MAX, UINT32_MAX };
        for (int i = 0; i < n; ++i) {
            if (sizes[i].width < s.width) s.width = sizes[i].width;
            if (sizes[i].height < s.height) s.height = sizes[i].height;
        }
        *out = s;
        return KC_OK;
    }
    case KC_POLICY_SPECIFIC_SLOT:
        *out = (slot_index >= 0 && slot_index < n) ? sizes[slot_index] : kc_size{ 1, 1 };
        return KC_OK;
    case KC_POLICY_SPECIFIC_SIZE:
        *out = specific;
        return KC_OK;
    }
    set_error("invalid ResizePolicy");
    return KC_ERR_INVALID_ARG;
}

// mix::process, src/node/mix.rs:51-134.
int mix_process(kc_image *left_in, kc_image *right_in, int mix_type, kc_image **out)
{
    KC_PROF("mix_process");
    *out = nullptr;
    kc_image *left = nullptr, *right = nullptr;
    if (left_in) {
        const bool rgba = left_in->is_rgba();
        if (right_in) {
            KC_TRY(image_as_type(right_in, rgba, &right));
        } else {
            KC_TRY(image_from_value(kc_size{ left_in->w(), left_in->h() }, 0.0f, rgba, &right));
        }
        left = left_in;
        image_retain(left);
    } else if (right_in) {
        KC_TRY(image_from_value(kc_size{ right_in->w(), right_in->h() }, 0.0f, right_in->is_rgba(), &left));
        right = right_in;
        image_retain(right);
    } else {
        return image_from_value(kc_size{ 1, 1 }, 0.0f, false, out);
    }
    int s = KC_OK;
    if (left->is_rgba() != right->is_rgba()) {
        // mix.rs:126 -> empty Vec (cannot happen after as_type, kept for fidelity)
    } else if (left->w() != right->w() || left->h() != right->h()) {
        set_error("Mix inputs differ in size; the resize pre-step (resize_buffers) must run first");
        s = KC_ERR_INVALID_ARG;
    } else if (left->is_rgba()) {
        // R, G, B mixed; A := 1.0; input alphas are never read (mix.rs:199-213)
        kc_plane *p[4] = { nullptr, nullptr, nullptr, nullptr };
        s = planes_mix_prepare(left->planes, right->planes, 3);
        for (int c = 0; c < 3 && s == KC_OK; ++c) s = plane_mix(mix_type, left->planes[c], right->planes[c], &p[c]);
        if (s == KC_OK) {
            p[3] = plane_new_const(left->w(), left->h(), 1.0f);
            *out = image_new(4, p);
        }
        for (int c = 0; c < 4; ++c) plane_release(p[c]);
    } else {
        kc_plane *p = nullptr;
        s = plane_mix(mix_type, left->planes[0], right->planes[0], &p);
        if (s == KC_OK) {
            *out = image_new(1, &p);
            plane_release(p);
        }
    }
    image_release(left);
    image_release(right);
    // Fusion off: every node materialises its planes -- R, G and B in ONE launch (blockIdx.y),
    // where the reference loops over them sequentially on one thread (mix.rs:199-213).
    if (s == KC_OK && *out && !ctx().fusion) {
        s = image_force(*out);
        if (s != KC_OK) {
            image_release(*out);
            *out = nullptr;
        }
    }
    return s;
}

// separate_rgba::process, src/node/separate_rgba.rs:38-69: four gray images aliasing the planes.
int separate_process(kc_image *in, kc_image *out[4])
{
    for (int i = 0; i < 4; ++i) {
        if (in && in->is_rgba()) {
            out[i] = image_new(1, &in->planes[i]);
        } else {
            kc_plane *z = plane_new_const(1, 1, 0.0f);  // default_output: pixel_buffer(0.0)
            out[i] = image_new(1, &z);
            plane_release(z);
        }
    }
    return KC_OK;
}

// combine_rgba::process, src/node/combine_rgba.rs:14-97.  in[slot] = input on that slot or NULL;
// the size of the defaults comes from slot_datas.get(0) = the lowest connected slot.
int combine_process(kc_image *const in[4], kc_image **out)
{
    kc_size size{ 1, 1 };
    for (int i = 0; i < 4; ++i)
        if (in[i]) {
            size = kc_size{ in[i]->w(), in[i]->h() };
            break;
        }
    kc_plane *zero = nullptr;  // one shared zero plane for missing R, G, B (:30-44)
    kc_plane *p[4];
    for (int i = 0; i < 4; ++i) {
        if (in[i]) {
            if (in[i]->is_rgba()) {
                set_error("It shouldn't be possible to connect an RGBA image into this slot");
                for (int j = 0; j < i; ++j) plane_release(p[j]);
                plane_release(zero);
                return KC_ERR_INVALID_SLOT_TYPE;
            }
            p[i] = in[i]->planes[0];
            plane_retain(p[i]);
        } else if (i == 3) {
            p[i] = plane_new_const(size.width, size.height, 1.0f);
        } else {
            if (!zero) zero = plane_new_const(size.width, size.height, 0.0f);
            p[i] = zero;
            plane_retain(p[i]);
        }
    }
    *out = image_new(4, p);
    for (int i = 0; i < 4; ++i) plane_release(p[i]);
    plane_release(zero);
    return KC_OK;
}

// value::process, src/node/value.rs:14-26: a 1x1 gray plane.
int value_process(float v, kc_image **out)
{
    kc_plane *p = plane_new_const(1, 1, v);
    *out = image_new(1, &p);
    plane_release(p);
    return KC_OK;
}

// height_to_normal::process, src/node/height_to_normal.rs:16-77.  full_h == 0: the whole image.  full_h > 0: `in` is a
// row band preceded by its halo row (rows + 1 rows) of an image full_h rows high; the result has `rows` rows.
static int height_to_normal_impl(kc_image *in, uint32_t full_h, kc_image **out)
{
    *out = nullptr;
    if (!in || in->is_rgba()) return KC_OK;  // reference returns an empty Vec
    KC_TRY(need_init());
    Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    kc_plane *src = in->planes[0];
    KC_TRY(plane_materialize(src));
    const bool band = full_h != 0;
    if (band && src->h < 2) {
        set_error("height_to_normal band: the input must hold the halo row and at least one row");
        return KC_ERR_INVALID_ARG;
    }
    const uint32_t w = src->w, h = band ? src->h - 1 : src->h;
    kc_plane *p[4] = { nullptr, nullptr, nullptr, nullptr };
    int s = KC_OK;
    for (int i = 0; i < 3 && s == KC_OK; ++i) s = plane_new_mem(w, h, &p[i]);
    if (s == KC_OK) {
        hipError_t e = launch_height_to_normal(src->dptr, (uint32_t)(src->pitch / 4), w, h, band ? full_h : h, band ? 1 : 0,
                                               p[0]->dptr, p[1]->dptr, p[2]->dptr, (uint32_t)(p[0]->pitch / 4),
                                               // three result planes against one input: once they do not all fit the
                                               // Infinity Cache the results are streamed (49.6 against 51.6 us at 4096^2)
                                               cache_policy_mask((uint64_t)w * h * 4, (uint64_t)w * h * 12, 1) ? 0x100u : 0u, c.stream);
        if (e != hipSuccess) s = hip_fail(e, "launch_height_to_normal");
        else {
            c.launches++;
            c.alg_bytes += (uint64_t)w * h * 16;  // 4 B read + 12 B written per pixel
        }
    }
    if (s == KC_OK) {
        p[3] = plane_new_const(w, h, 1.0f);  // from_buffers_rgb appends a ones plane (slot_image.rs:90-102)
        *out = image_new(4, p);
    }
    for (int i = 0; i < 4; ++i) plane_release(p[i]);
    return s;
}

int height_to_normal_process(kc_image *in, kc_image **out) { return height_to_normal_impl(in, 0, out); }

int height_to_normal_band(kc_image *in_with_halo, uint32_t full_h, kc_image **out)
{
    if (full_h == 0) {
        set_error("height_to_normal band: full height must be given");
        return KC_ERR_INVALID_ARG;
    }
    return height_to_normal_impl(in_with_halo, full_h, out);
}

}  // namespace kc
