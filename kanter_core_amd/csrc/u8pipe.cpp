// The host boundary as a pipeline: interleaved u8 images in, interleaved RGBA8 images out, with the transfers of one image
// overlapping the evaluation of another.
//
// Every real use of the reference enters through deconstruct_image (src/shared.rs:16-56: u8 -> planar f32, /255.) and leaves
// through SlotImage::to_u8 / to_u8_srgb (src/slot_image.rs:141-207).  kc_image_from_u8 / kc_image_to_u8 are those two as blocking
// calls on pageable memory: the copy and the kernels of one image run back to back on the one compute stream and the host waits
// for them, so PCIe idles while the GPU computes and the GPU idles while PCIe copies.  A batch job (one graph over many images)
// wants the opposite, and the u8 route moves a quarter of the bytes of the f32 one (4 B per RGBA pixel each way instead of 16):
//
//   kc_u8_pipe        `depth` slots; each slot owns a PINNED host buffer for an input image, a pinned host buffer for an output
//                     image and a device staging buffer for each.  Two copy streams (up, down) beside the compute stream.
//   upload(slot)      H2D of the slot's input buffer on the UP stream (behind the event that says the slot's previous image has
//                     been converted), then -- on the compute stream, behind the copy's event -- from_u8_kernel into fresh planes.
//                     Returns the image at once; nothing waits on the host.  Called for image k + 1 after the evaluation of image k
//                     has been enqueued, the copy runs during that evaluation.
//   download(slot)    to_u8_kernel of the result on the compute stream into the slot's device staging (behind the event that
//                     says the slot's previous download has left it), then D2H on the DOWN stream behind the kernel's event.
//   wait_download     blocks until the slot's D2H has finished: the only host-side wait of the loop, one or more images behind the
//                     evaluation being enqueued.
// The staging buffers are the pipe's own (not pool blocks): a pool block is recycled in compute-stream order, and waiting for
// that order on the copy streams would serialise exactly what this is meant to overlap.
#include "kc_runtime.hpp"

struct kc_u8_pipe {
    uint32_t w = 0, h = 0;
    int channels = 4, depth = 0;
    size_t in_bytes = 0, out_bytes = 0;
    hipStream_t up = nullptr, down = nullptr;
    struct Slot {
        uint8_t *host_in = nullptr, *host_out = nullptr;  // pinned
        uint8_t *dev_in = nullptr, *dev_out = nullptr;
        hipEvent_t copied_in = nullptr;   // up stream: the input image is in dev_in
        hipEvent_t converted = nullptr;   // compute stream: from_u8 has read dev_in (the next upload may overwrite it)
        hipEvent_t packed = nullptr;      // compute stream: to_u8 has filled dev_out
        hipEvent_t copied_out = nullptr;  // down stream: host_out holds the image (the next to_u8 may overwrite dev_out)
        bool in_used = false, out_used = false;
    };
    std::vector<Slot> slots;
};

namespace kc {

int u8_pipe_free(kc_u8_pipe *p)
{
    if (!p) return KC_OK;
    if (p->up) (void)hipStreamSynchronize(p->up);
    if (p->down) (void)hipStreamSynchronize(p->down);
    if (ctx().stream) (void)hipStreamSynchronize(ctx().stream);
    for (auto &s : p->slots) {
        if (s.host_in) (void)hipHostFree(s.host_in);
        if (s.host_out) (void)hipHostFree(s.host_out);
        if (s.dev_in) (void)hipFree(s.dev_in);
        if (s.dev_out) (void)hipFree(s.dev_out);
        for (hipEvent_t e : { s.copied_in, s.converted, s.packed, s.copied_out })
            if (e) (void)hipEventDestroy(e);
    }
    if (p->up) (void)hipStreamDestroy(p->up);
    if (p->down) (void)hipStreamDestroy(p->down);
    delete p;
    return KC_OK;
}

int u8_pipe_create(uint32_t w, uint32_t h, int channels, int depth, kc_u8_pipe **out)
{
    KC_TRY(need_init());
    if (w == 0 || h == 0 || channels < 1 || channels > 4 || depth < 1 || depth > 16 || !out) {
        set_error("kc_u8_pipe_create: width and height > 0, 1..4 channels, 1..16 slots");
        return KC_ERR_INVALID_ARG;
    }
    kc_u8_pipe *p = new kc_u8_pipe();
    p->w = w;
    p->h = h;
    p->channels = channels;
    p->depth = depth;
    p->in_bytes = (size_t)w * h * channels;
    p->out_bytes = (size_t)w * h * 4;
    p->slots.resize((size_t)depth);
    hipError_t e = hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking);
    for (auto &s : p->slots) {
        if (e == hipSuccess) e = hipHostMalloc((void **)&s.host_in, p->in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s.host_out, p->out_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s.dev_in, (p->in_bytes + 255) / 256 * 256);
        if (e == hipSuccess) e = hipMalloc((void **)&s.dev_out, (p->out_bytes + 255) / 256 * 256);
        for (hipEvent_t *ev : { &s.copied_in, &s.converted, &s.packed, &s.copied_out })
            if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        const int st = e == hipErrorOutOfMemory ? KC_ERR_OUT_OF_MEMORY : hip_fail(e, "kc_u8_pipe_create");
        if (st == KC_ERR_OUT_OF_MEMORY) set_error("kc_u8_pipe_create: out of (pinned or device) memory");
        (void)hipGetLastError();
        (void)u8_pipe_free(p);
        return st;
    }
    *out = p;
    return KC_OK;
}

static int slot_of(kc_u8_pipe *p, int slot, kc_u8_pipe::Slot **s)
{
    if (!p || slot < 0 || slot >= p->depth) {
        set_error("kc_u8_pipe: no such slot");
        return KC_ERR_INVALID_ARG;
    }
    *s = &p->slots[(size_t)slot];
    return KC_OK;
}

int u8_pipe_buffers(kc_u8_pipe *p, int slot, uint8_t **host_in, const uint8_t **host_out)
{
    kc_u8_pipe::Slot *s = nullptr;
    KC_TRY(slot_of(p, slot, &s));
    if (host_in) *host_in = s->host_in;
    if (host_out) *host_out = s->host_out;
    return KC_OK;
}

// deconstruct_image (src/shared.rs:16-56) of the slot's input buffer, asynchronously.
int u8_pipe_upload(kc_u8_pipe *p, int slot, kc_image **out)
{
    kc_u8_pipe::Slot *s = nullptr;
    KC_TRY(slot_of(p, slot, &s));
    Context &c = ctx();
    kc_plane *pl[4] = { nullptr, nullptr, nullptr, nullptr };
    float *dp[4] = { nullptr, nullptr, nullptr, nullptr };
    int st = KC_OK;
    for (int i = 0; i < 4 && st == KC_OK; ++i) {
        if (i < p->channels) {
            st = plane_new_mem(p->w, p->h, &pl[i]);
            if (st == KC_OK) dp[i] = pl[i]->dptr;
        } else {
            pl[i] = plane_new_const(p->w, p->h, i == 3 ? 1.0f : 0.0f);
        }
    }
    if (st == KC_OK) {
        hipError_t e = hipSuccess;
        if (s->in_used) e = hipStreamWaitEvent(p->up, s->converted, 0);  // the slot's previous image has been read
        if (e == hipSuccess) e = hipMemcpyAsync(s->dev_in, s->host_in, p->in_bytes, hipMemcpyHostToDevice, p->up);
        if (e == hipSuccess) e = hipEventRecord(s->copied_in, p->up);
        if (e == hipSuccess) e = hipStreamWaitEvent(c.stream, s->copied_in, 0);
        if (e == hipSuccess)
            e = launch_from_u8(s->dev_in, p->channels, p->w, p->h, dp, (uint32_t)(pl[0]->pitch / 4),
                               cache_policy_mask(p->in_bytes, (uint64_t)p->w * p->h * 4 * p->channels, 1), c.stream);
        if (e == hipSuccess) e = hipEventRecord(s->converted, c.stream);
        if (e != hipSuccess) st = hip_fail(e, "kc_u8_pipe_upload");
        else {
            s->in_used = true;
            c.launches++;
            c.alg_bytes += (uint64_t)p->w * p->h * p->channels * 5;  // 1 B read + 4 B written per sample
        }
    }
    if (st == KC_OK) *out = image_new(4, pl);
    for (int i = 0; i < 4; ++i) plane_release(pl[i]);
    return st;
}

// SlotImage::to_u8 / to_u8_srgb (src/slot_image.rs:141-207) into the slot's output buffer, asynchronously.
int u8_pipe_download(kc_u8_pipe *p, int slot, kc_image *img, bool srgb)
{
    kc_u8_pipe::Slot *s = nullptr;
    KC_TRY(slot_of(p, slot, &s));
    if (!img || img->w() != p->w || img->h() != p->h) {
        set_error("kc_u8_pipe_download: the image has another size than the pipe's");
        return KC_ERR_INVALID_ARG;
    }
    Context &c = ctx();
    KC_TRY(image_force(img));
    Operand o[4];
    for (int i = 0; i < 4; ++i) o[i] = plane_operand(img->planes[img->is_rgba() ? i : 0]);
    uint32_t n_res = 0;
    for (int i = 0; i < (img->is_rgba() ? 4 : 1); ++i) n_res += o[i].ptr != nullptr;
    hipError_t e = hipSuccess;
    if (s->out_used) e = hipStreamWaitEvent(c.stream, s->copied_out, 0);  // the slot's previous image has left dev_out
    if (e == hipSuccess)
        e = launch_to_u8(o[0], o[1], o[2], o[3], img->is_rgba() ? 0 : 1, srgb ? 1 : 0, p->w, p->h, s->dev_out,
                         cache_policy_mask((uint64_t)p->w * p->h * 4 * n_res, p->out_bytes, n_res ? n_res : 1), c.stream);
    if (e == hipSuccess) e = hipEventRecord(s->packed, c.stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(p->down, s->packed, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(s->host_out, s->dev_out, p->out_bytes, hipMemcpyDeviceToHost, p->down);
    if (e == hipSuccess) e = hipEventRecord(s->copied_out, p->down);
    if (e != hipSuccess) return hip_fail(e, "kc_u8_pipe_download");
    s->out_used = true;
    c.launches++;
    c.alg_bytes += (uint64_t)p->w * p->h * 4 * (n_res + 1);
    return KC_OK;
}

int u8_pipe_wait_download(kc_u8_pipe *p, int slot)
{
    kc_u8_pipe::Slot *s = nullptr;
    KC_TRY(slot_of(p, slot, &s));
    if (!s->out_used) return KC_OK;
    KC_HIP(hipEventSynchronize(s->copied_out));
    return KC_OK;
}

}  // namespace kc
