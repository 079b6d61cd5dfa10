// extern "C" surface declared in include/kanter_core_amd.h.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "kc_runtime.hpp"

namespace kc {
const std::string &last_error();
}
using namespace kc;

#define KC_ARG(cond)                                         \
    do {                                                     \
        if (!(cond)) {                                       \
            set_error("invalid argument: " #cond);           \
            return KC_ERR_INVALID_ARG;                       \
        }                                                    \
    } while (0)

typedef std::lock_guard<std::recursive_mutex> Lock;

// No C++ exception may cross the C ABI (callers are C, Rust, ctypes): every entry point that can
// allocate is a function-try-block ending in KC_CATCH.
#define KC_CATCH                                                        \
    catch (const std::bad_alloc &)                                      \
    {                                                                   \
        set_error("host allocation failed");                            \
        return KC_ERR_OUT_OF_MEMORY;                                    \
    }                                                                   \
    catch (const std::exception &kc_exc_)                                   \
    {                                                                   \
        set_error(std::string("internal error: ") + kc_exc_.what());        \
        return KC_ERR_GENERIC;                                          \
    }                                                                   \
    catch (...)                                                         \
    {                                                                   \
        set_error("internal error");                                    \
        return KC_ERR_GENERIC;                                          \
    }

extern "C" {

// ---------------------------------------------------------------- context
int kc_init(int device_ordinal)
try {
    Context &c = ctx();
    Lock lk(c.mu);
    if (c.inited) {
        if (c.device == device_ordinal) return KC_OK;
        set_error("kc_init: already bound to another device (one process per GPU)");
        return KC_ERR_INVALID_ARG;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available: kanter_core_amd has no CPU fallback");
        return KC_ERR_NO_DEVICE;
    }
    KC_ARG(device_ordinal >= 0 && device_ordinal < count);
    KC_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    KC_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        return KC_ERR_NO_DEVICE;
    }
    KC_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    c.stream = c.own_stream;
    c.device = device_ordinal;
    if (const char *mb = std::getenv("KC_MAX_BLOCKS")) {
        int v = std::atoi(mb);
        if (v >= 1) c.max_blocks = v;
    }
    if (const char *rm = std::getenv("KC_RESIZE_MODE")) c.resize_mode = std::atoi(rm);
    if (const char *cp = std::getenv("KC_CACHE_POLICY")) c.cache_policy = std::atoi(cp) != 0;
    if (const char *cb = std::getenv("KC_CACHE_BUDGET_MB")) c.cache_budget_mb = std::max(0, std::atoi(cb));
    if (const char *c1 = std::getenv("KC_CHAIN1")) c.chain1 = std::atoi(c1) != 0;
    if (const char *j = std::getenv("KC_JOIN")) c.join = std::atoi(j) != 0;
    if (const char *wd = std::getenv("KC_WIDE")) c.wide = std::atoi(wd) != 0;
    if (const char *d2 = std::getenv("KC_DOWN2")) c.down2 = std::max(0, std::min(2, std::atoi(d2)));
    if (const char *p2 = std::getenv("KC_POLY2")) c.poly2 = std::atoi(p2) != 0;
    if (const char *br = std::getenv("KC_DOWN2_BY_ROWS")) c.down2_by_rows = std::max(-1, std::min(1, std::atoi(br)));
    if (const char *p2r = std::getenv("KC_POLY2_MIN_RATIO")) c.poly2_min_ratio = std::max(2, std::atoi(p2r));
    if (const char *rt = std::getenv("KC_RESIZE_TILE_H")) c.resize_tile_h = std::atoi(rt);
    if (const char *rw = std::getenv("KC_RESIZE_TILE_W")) c.resize_tile_w = std::atoi(rw);
    if (const char *cu = std::getenv("KC_CHAIN_UNROLL")) {
        int v = std::atoi(cu);
        if (v == 1 || v == 2 || v == 4 || v == 6 || v == 8) c.chain_unroll = v;
    }
    if (const char *sp = std::getenv("KC_SPECIALIZE")) {
        int v = std::atoi(sp);
        if (v >= 0 && v <= 2) specialize_set_mode(v, 0);
    }
    c.inited = true;
#ifdef KC_HOST_SAMPLE
    sampler_start();
#endif
    return KC_OK;
}
KC_CATCH

int kc_shutdown(void)
try {
    Context &c = ctx();
    Lock lk(c.mu);
    if (!c.inited) return KC_OK;
#ifdef KC_HOST_PROFILE
    prof_report();
#endif
#ifdef KC_HOST_SAMPLE
    sampler_report();
#endif
    (void)hipStreamSynchronize(c.stream);
    (void)comm_destroy();
    specialize_shutdown();
    pool_trim();
    for (auto &us : c.upload_ring) {
        if (us.copied) (void)hipEventDestroy(us.copied);
        if (us.host) (void)hipHostFree(us.host);
        us = Context::UploadSlot{};
    }
    c.upload_next = 0;
    for (auto &kv : c.taps) (void)hipFree(kv.second.dev_block);
    c.taps.clear();
    for (auto &kv : c.band_taps) (void)hipFree(kv.second.dev_block);
    c.band_taps.clear();
    if (c.own_stream) (void)hipStreamDestroy(c.own_stream);
    c.own_stream = c.stream = nullptr;
    c.inited = false;
    c.device = -1;
    return KC_OK;
}
KC_CATCH

int kc_is_initialized(void) { return ctx().inited ? 1 : 0; }

int kc_set_stream(void *hip_stream)
try {
    KC_TRY(need_init());
    Context &c = ctx();
    Lock lk(c.mu);
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : c.own_stream;
    if (next != c.stream) {
        // pool blocks are recycled in stream order: drain the old stream before switching
        KC_HIP(hipStreamSynchronize(c.stream));
        c.stream = next;
    }
    return KC_OK;
}
KC_CATCH

void *kc_get_stream(void) { return (void *)ctx().stream; }

int kc_sync(void)
try {
    KC_TRY(need_init());
    KC_HIP(hipStreamSynchronize(ctx().stream));
    {
        Lock lk(ctx().mu);
        comm_sync();  // sends still in flight hold references: let them go
    }
    return KC_OK;
}
KC_CATCH

const char *kc_last_error(void) { return last_error().c_str(); }

const char *kc_status_string(int s)
{
    switch (s) {
    case KC_OK: return "ok";
    case KC_ERR_GENERIC: return "Something went wrong";
    case KC_ERR_CANCELED: return "Node processing was canceled";
    case KC_ERR_IMAGE: return "Image error";
    case KC_ERR_INVALID_BUFFER_COUNT: return "Invalid number of channels";
    case KC_ERR_INVALID_NODE_ID: return "Invalid `NodeId`";
    case KC_ERR_INVALID_NODE_TYPE: return "Invalid `NodeType`";
    case KC_ERR_INVALID_SLOT_ID: return "Invalid `SlotId`";
    case KC_ERR_INVALID_SLOT_TYPE: return "Invalid `SlotType`";
    case KC_ERR_INVALID_EDGE: return "Invalid `Edge`";
    case KC_ERR_NO_SLOT_DATA: return "Could not find a `SlotData`";
    case KC_ERR_SLOT_OCCUPIED: return "`SlotId` is already in use";
    case KC_ERR_SLOT_NOT_OCCUPIED: return "`SlotId` is not in use";
    case KC_ERR_UNABLE_TO_LOCK: return "Unable to get a lock";
    case KC_ERR_NODE_PROCESSING: return "Error during node processing";
    case KC_ERR_POISON: return "Error with poisoned lock";
    case KC_ERR_TRY_LOCK: return "Error when trying to lock";
    case KC_ERR_NODE_DIRTY: return "The node is not up to date";
    case KC_ERR_IO: return "I/O error";
    case KC_ERR_INVALID_NAME: return "Invalid name, can only contain lowercase letters, numbers and underscores";
    case KC_ERR_HIP: return "HIP runtime error";
    case KC_ERR_NO_DEVICE: return "no gfx950 device (no CPU fallback)";
    case KC_ERR_INVALID_ARG: return "invalid argument";
    case KC_ERR_OUT_OF_MEMORY: return "out of HBM";
    case KC_ERR_UNSUPPORTED: return "unsupported";
    }
    return "unknown status";
}

int kc_set_fusion(int enabled)
try {
    ctx().fusion = enabled != 0;
    return KC_OK;
}
KC_CATCH

int kc_get_fusion(void) { return ctx().fusion ? 1 : 0; }

int kc_set_resize_mode(int mode)
try {
    KC_ARG(mode >= 0 && mode <= 4);
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    ctx().resize_mode = mode;
    return KC_OK;
}
KC_CATCH

int kc_get_resize_mode(void) { return ctx().resize_mode; }

int kc_set_cache_policy(int mode)
try {
    KC_ARG(mode == 0 || mode == 1);
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    ctx().cache_policy = mode;
    return KC_OK;
}
KC_CATCH

int kc_get_cache_policy(void) { return ctx().cache_policy; }

int kc_set_option(const char *name, int value)
try {
    KC_ARG(name);
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    if (std::strcmp(name, "chain1") == 0) ctx().chain1 = value != 0;
    else if (std::strcmp(name, "replay") == 0) ctx().replay = value != 0;
    else if (std::strcmp(name, "join") == 0) ctx().join = value != 0;
    else if (std::strcmp(name, "wide") == 0) ctx().wide = value != 0;
    else if (std::strcmp(name, "down2") == 0 && value >= 0 && value <= 2) ctx().down2 = value;
    else if (std::strcmp(name, "down2_by_rows") == 0 && value >= -1 && value <= 1) ctx().down2_by_rows = value;
    else if (std::strcmp(name, "poly2") == 0 && value >= 0 && value <= 1) ctx().poly2 = value;
    else if (std::strcmp(name, "poly2_min_ratio") == 0 && value >= 2) ctx().poly2_min_ratio = value;
    else if (std::strcmp(name, "cache_budget_mb") == 0 && value >= 0) ctx().cache_budget_mb = value;
    else if (std::strcmp(name, "link_gbps") == 0 && value > 0) ctx().link_gbps = value;
    else if (std::strcmp(name, "hbm_gbps") == 0 && value > 0) ctx().hbm_gbps = value;
    else {
        set_error(std::string("unknown option ") + name);
        return KC_ERR_INVALID_ARG;
    }
    return KC_OK;
}
KC_CATCH

int kc_get_option(const char *name, int *value)
try {
    KC_ARG(name && value);
    if (std::strcmp(name, "chain1") == 0) *value = ctx().chain1 ? 1 : 0;
    else if (std::strcmp(name, "replay") == 0) *value = ctx().replay ? 1 : 0;
    else if (std::strcmp(name, "join") == 0) *value = ctx().join ? 1 : 0;
    else if (std::strcmp(name, "wide") == 0) *value = ctx().wide ? 1 : 0;
    else if (std::strcmp(name, "down2") == 0) *value = ctx().down2;
    else if (std::strcmp(name, "down2_by_rows") == 0) *value = ctx().down2_by_rows;
    else if (std::strcmp(name, "poly2") == 0) *value = ctx().poly2;
    else if (std::strcmp(name, "poly2_min_ratio") == 0) *value = ctx().poly2_min_ratio;
    else if (std::strcmp(name, "cache_budget_mb") == 0) *value = ctx().cache_budget_mb;
    else if (std::strcmp(name, "link_gbps") == 0) *value = ctx().link_gbps;
    else if (std::strcmp(name, "hbm_gbps") == 0) *value = ctx().hbm_gbps;
    else {
        set_error(std::string("unknown option ") + name);
        return KC_ERR_INVALID_ARG;
    }
    return KC_OK;
}
KC_CATCH

int kc_resize_upsample_plan(uint32_t in_n, uint32_t out_n, int filter, int *eligible, int32_t info[5], float *rows, size_t cap)
try {
    KC_ARG(eligible && info);
    TapsHost t;
    KC_TRY(build_taps_host(in_n, out_n, filter, t));
    *eligible = t.up_ok ? 1 : 0;
    if (!t.up_ok) return KC_OK;
    info[0] = (int32_t)t.up.ratio;
    info[1] = (int32_t)t.up.taps;
    info[2] = t.up.off;
    info[3] = (int32_t)t.up.b_lo;
    info[4] = (int32_t)t.up.b_hi;
    if (rows)
        for (size_t i = 0; i < t.up_rows.size() && i < cap; ++i) rows[i] = t.up_rows[i];
    return KC_OK;
}
KC_CATCH

int kc_resize_down2_plan(uint32_t in_n, uint32_t out_n, int filter, int32_t info[5], uint32_t *left_count, float *w, size_t wcap,
                         uint32_t *vrec, size_t vcap, float *hw, size_t hcap)
try {
    KC_ARG(info);
    TapsHost t;
    KC_TRY(build_taps_host(in_n, out_n, filter, t));
    info[0] = (int32_t)t.stride;
    info[1] = (int32_t)t.d2_nc;
    info[2] = (int32_t)t.d2_hstride;
    info[3] = (int32_t)t.d2_tile_w;
    info[4] = (int32_t)t.min_count;
    if (left_count)
        for (uint32_t o = 0; o < out_n; ++o) {
            left_count[o] = t.left[o];
            left_count[out_n + o] = t.count[o];
        }
    if (w)
        for (size_t i = 0; i < t.w.size() && i < wcap; ++i) w[i] = t.w[i];
    if (vrec)
        for (size_t i = 0; i < t.d2_vrec.size() && i < vcap; ++i) vrec[i] = t.d2_vrec[i];
    if (hw)
        for (size_t i = 0; i < t.d2_hw.size() && i < hcap; ++i) hw[i] = t.d2_hw[i];
    return KC_OK;
}
KC_CATCH

int kc_set_specialize(int mode, int after)
try {
    KC_ARG(mode >= 0 && mode <= 2);
    return specialize_set_mode(mode, after);
}
KC_CATCH

int kc_get_specialize(void) { return specialize_get_mode(); }

int kc_specialize_wait(void)
try {
    specialize_wait();
    return KC_OK;
}
KC_CATCH

int kc_specialize_stats(uint64_t *compiled, uint64_t *failed, uint64_t *launches, uint64_t *pending)
try {
    specialize_stats(compiled, failed, launches, pending);
    return KC_OK;
}
KC_CATCH

int kc_specialize_compile_check(const uint32_t *words, uint32_t n_ops, uint32_t n_in, int start_src, int flat, char *source,
                                size_t cap)
try {
    KC_ARG(words && n_ops >= 1 && n_ops <= (uint32_t)KC_CHAIN_MAX_OPS && n_in <= (uint32_t)KC_CHAIN_MAX_IN);
    KC_ARG(start_src >= -1 && start_src < (int)n_in);
    ChainProgram P;
    std::memset(&P, 0, sizeof P);
    P.n_ops = n_ops;
    P.n_in = n_in;
    P.start_src = start_src;
    P.rows = flat ? 1u : 2u;
    P.row_units = 1;
    for (uint32_t i = 0; i < n_ops; ++i) {
        const uint32_t from = (words[i] >> 8) & 0xffu, level = words[i] >> 16;
        KC_ARG((words[i] & 0xffu) <= CH_SAVE_LOAD && (from <= n_in || (from == (uint32_t)KC_CHAIN_SRC_SAVED && (words[i] & 0xffu) != CH_SAVE_LOAD)));
        KC_ARG(level < (uint32_t)KC_CHAIN_MAX_SAVED && (level == 0 || from == (uint32_t)KC_CHAIN_SRC_SAVED || (words[i] & 0xffu) == CH_SAVE_LOAD));
        ((i & 1u) ? P.step[0][i / 2].b : P.step[0][i / 2].a).word = words[i];
    }
    if (source && cap) {
        const std::string src = specialize_source(P);
        std::snprintf(source, cap, "%s", src.c_str());
    }
    std::string log;
    int s = specialize_compile_only(P, &log);
    if (s != KC_OK) set_error("specialised chain kernel did not compile: " + log);
    return s;
}
KC_CATCH

int kc_kernel_cache_set_dir(const char *dir)
try {
    Lock lk(ctx().mu);
    return kernel_cache_set_dir(dir);
}
KC_CATCH

int kc_kernel_cache_stats(uint64_t *hits, uint64_t *rejected, uint64_t *written, uint64_t *kernels_loaded)
try {
    kernel_cache_stats(hits, rejected, written, kernels_loaded);
    return KC_OK;
}
KC_CATCH

int kc_kernel_cache_precompile(const uint32_t *words, uint32_t n_ops, uint32_t n_in, int start_src, int flat, uint32_t nt_mask,
                               uint32_t up_taps, int up_wide, const char *dir)
try {
    KC_ARG(words && dir);
    std::string log;
    const int s = kernel_cache_precompile(words, n_ops, n_in, start_src, flat != 0, nt_mask, up_taps, up_wide != 0, dir, &log);
    if (s != KC_OK) set_error("kernel cache: " + log);
    return s;
}
KC_CATCH

// Forgets every kernel this process has compiled or loaded (the files stay): the next sighting of a program is a first one.
int kc_specialize_reset(void)
try {
    Lock lk(ctx().mu);
    if (ctx().stream) (void)hipStreamSynchronize(ctx().stream);
    const int mode = specialize_get_mode();
    specialize_shutdown();
    (void)mode;
    return KC_OK;
}
KC_CATCH

int kc_specialize_compile_check_upsample(const uint32_t *words, uint32_t n_ops, uint32_t n_in, int start_src, uint32_t taps, int wide,
                                         char *source, size_t cap)
try {
    KC_ARG(words && n_ops >= 1 && n_ops <= (uint32_t)KC_CHAIN_MAX_OPS && n_in >= 1 && n_in <= (uint32_t)KC_CHAIN_INTERP_IN);
    KC_ARG(start_src >= -1 && start_src < (int)n_in && (taps == 1 || taps == 3));
    ChainProgram P;
    std::memset(&P, 0, sizeof P);
    P.n_ops = n_ops;
    P.n_in = n_in;
    P.start_src = start_src;
    for (uint32_t i = 0; i < n_ops; ++i) {
        const uint32_t from = words[i] >> 8, code = words[i] & 0xffu;
        KC_ARG(code <= CH_MUL_INV && from <= n_in && code != CH_DIV_L && code != CH_DIV_R && code != CH_POW_L && code != CH_POW_R);
        ((i & 1u) ? P.step[0][i / 2].b : P.step[0][i / 2].a).word = words[i];
    }
    UpsampleArgs U{};
    U.H.taps = U.V.taps = taps;
    U.tile_w = wide ? 1024u : 256u;
    if (source && cap) {
        const std::string src = specialize_source(P, &U);
        std::snprintf(source, cap, "%s", src.c_str());
    }
    std::string log;
    int s = specialize_compile_only(P, &log, &U);
    if (s != KC_OK) set_error("specialised up-sampling kernel did not compile: " + log);
    return s;
}
KC_CATCH

int kc_stats(uint64_t *in_use, uint64_t *cached, uint64_t *launches)
try {
    Context &c = ctx();
    Lock lk(c.mu);
    if (in_use) *in_use = c.bytes_in_use;
    if (cached) *cached = c.bytes_cached;
    if (launches) *launches = c.launches;
    return KC_OK;
}
KC_CATCH

int kc_stats_counter(const char *name, uint64_t *value)
try {
    KC_ARG(name && value);
    Context &c = ctx();
    Lock lk(c.mu);
    auto it = c.counters.find(name);
    *value = it == c.counters.end() ? 0 : it->second;
    return KC_OK;
}
KC_CATCH

int kc_stats_algorithmic_bytes(uint64_t *bytes)
try {
    KC_ARG(bytes);
    Context &c = ctx();
    Lock lk(c.mu);
    *bytes = c.alg_bytes;
    return KC_OK;
}
KC_CATCH

int kc_pool_trim(void)
try {
    KC_TRY(need_init());
    return pool_trim();
}
KC_CATCH

// ---------------------------------------------------------------- planes
int kc_plane_alloc(uint32_t w, uint32_t h, kc_plane **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(out);
    return plane_new_mem(w, h, out);
}
KC_CATCH

int kc_plane_const(uint32_t w, uint32_t h, float v, kc_plane **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(out && w > 0 && h > 0);
    *out = plane_new_const(w, h, v);
    return KC_OK;
}
KC_CATCH

int kc_plane_wrap(void *dptr, uint32_t w, uint32_t h, size_t pitch, kc_plane **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_TRY(need_init());
    KC_ARG(out && dptr && w > 0 && h > 0);
    // kernels move 16 bytes per lane: rows must start 16-byte aligned and be readable in whole
    // float4 units (pitch covers the width rounded up to 4 floats)
    if (((uintptr_t)dptr & 15) || (pitch & 15) || pitch < ((size_t)(w + 3) / 4) * 16) {
        set_error("kc_plane_wrap: pointer and pitch must be 16-byte aligned and pitch >= 16*ceil(width/4)");
        return KC_ERR_INVALID_ARG;
    }
    kc_plane *p = new kc_plane();
    p->w = w;
    p->h = h;
    p->kind = kc_plane::MEM;
    p->dptr = (float *)dptr;
    p->pitch = pitch;
    p->owned = false;
    *out = p;
    return KC_OK;
}
KC_CATCH

int kc_plane_retain(kc_plane *p)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(p);
    plane_retain(p);
    return KC_OK;
}
KC_CATCH

int kc_plane_release(kc_plane *p)
try {
    if (!p) return KC_OK;
    Lock lk(ctx().mu);
    plane_release(p);
    return KC_OK;
}
KC_CATCH

int kc_plane_size(const kc_plane *p, uint32_t *w, uint32_t *h)
try {
    KC_ARG(p);
    if (w) *w = p->w;
    if (h) *h = p->h;
    return KC_OK;
}
KC_CATCH

int kc_plane_is_const(const kc_plane *p, int *is_const, float *v)
try {
    KC_ARG(p);
    if (is_const) *is_const = p->kind == kc_plane::CONST;
    if (v) *v = p->cval;
    return KC_OK;
}
KC_CATCH

int kc_plane_materialize(kc_plane *p)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(p);
    return plane_materialize(p);
}
KC_CATCH

int kc_plane_device_ptr(kc_plane *p, void **dptr, size_t *pitch)
try {
    KC_ARG(p);
    KC_TRY(plane_materialize(p));
    if (dptr) *dptr = p->dptr;
    if (pitch) *pitch = p->pitch;
    return KC_OK;
}
KC_CATCH

int kc_plane_upload_f32(kc_plane *p, const float *host, size_t host_pitch)
try {
    KC_TRY(need_init());
    KC_ARG(p && host && p->kind == kc_plane::MEM);
    if (host_pitch == 0) host_pitch = (size_t)p->w * 4;
    Lock lk(ctx().mu);
    KC_HIP(hipMemcpy2DAsync(p->dptr, p->pitch, host, host_pitch, (size_t)p->w * 4, p->h, hipMemcpyHostToDevice, ctx().stream));
    KC_HIP(hipStreamSynchronize(ctx().stream));
    return KC_OK;
}
KC_CATCH

int kc_plane_download_f32(kc_plane *p, float *host, size_t host_pitch)
try {
    KC_ARG(p && host);
    if (host_pitch == 0) host_pitch = (size_t)p->w * 4;
    Lock lk(ctx().mu);
    if (p->kind == kc_plane::CONST) {
        for (uint32_t y = 0; y < p->h; ++y) {
            float *row = (float *)((char *)host + y * host_pitch);
            for (uint32_t x = 0; x < p->w; ++x) row[x] = p->cval;
        }
        return KC_OK;
    }
    KC_TRY(need_init());
    KC_TRY(plane_force(p));
    KC_HIP(hipMemcpy2DAsync(host, host_pitch, p->dptr, p->pitch, (size_t)p->w * 4, p->h, hipMemcpyDeviceToHost, ctx().stream));
    KC_HIP(hipStreamSynchronize(ctx().stream));
    return KC_OK;
}
KC_CATCH

// ---------------------------------------------------------------- images
int kc_image_gray(kc_plane *p, kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(p && out);
    *out = image_new(1, &p);
    return KC_OK;
}
KC_CATCH

int kc_image_rgba(kc_plane *const planes[4], kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(planes && out && planes[0] && planes[1] && planes[2] && planes[3]);
    for (int i = 1; i < 4; ++i) KC_ARG(planes[i]->w == planes[0]->w && planes[i]->h == planes[0]->h);
    *out = image_new(4, planes);
    return KC_OK;
}
KC_CATCH

int kc_image_retain(kc_image *img)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(img);
    image_retain(img);
    return KC_OK;
}
KC_CATCH

int kc_image_release(kc_image *img)
try {
    if (!img) return KC_OK;
    Lock lk(ctx().mu);
    image_release(img);
    return KC_OK;
}
KC_CATCH

int kc_image_is_rgba(const kc_image *img, int *is_rgba)
try {
    KC_ARG(img && is_rgba);
    *is_rgba = img->is_rgba();
    return KC_OK;
}
KC_CATCH

int kc_image_size(const kc_image *img, kc_size *size)
try {
    KC_ARG(img && size);
    *size = kc_size{ img->w(), img->h() };
    return KC_OK;
}
KC_CATCH

int kc_image_plane(const kc_image *img, int channel, kc_plane **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(img && out && channel >= 0 && channel < img->n);
    *out = img->planes[channel];
    plane_retain(*out);
    return KC_OK;
}
KC_CATCH

int kc_image_from_value(kc_size size, float v, int rgba, kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(out);
    return image_from_value(size, v, rgba != 0, out);
}
KC_CATCH

int kc_image_as_type(const kc_image *img, int rgba, kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(img && out);
    return image_as_type(const_cast<kc_image *>(img), rgba != 0, out);
}
KC_CATCH

int kc_image_materialize(kc_image *img)
try {
    KC_ARG(img);
    Lock lk(ctx().mu);
    KC_TRY(image_force(img));
    for (int i = 0; i < img->n; ++i) KC_TRY(plane_materialize(img->planes[i]));
    return KC_OK;
}
KC_CATCH

int kc_image_from_u8(const uint8_t *host, uint32_t w, uint32_t h, int channels, kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(out);
    return image_from_u8(host, w, h, channels, out);  // synchronises: host buffer may be reused on return
}
KC_CATCH

int kc_image_to_u8(kc_image *img, int srgb, uint8_t *host)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(img && host);
    return image_to_u8(img, srgb != 0, host);
}
KC_CATCH

int kc_image_from_f32(const float *const host_planes[], int n, uint32_t w, uint32_t h, kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(host_planes && out && (n == 1 || n == 4));
    kc_plane *p[4] = { nullptr, nullptr, nullptr, nullptr };
    int s = KC_OK;
    for (int i = 0; i < n && s == KC_OK; ++i) {
        s = plane_new_mem(w, h, &p[i]);
        if (s == KC_OK) s = kc_plane_upload_f32(p[i], host_planes[i], 0);
    }
    if (s == KC_OK) *out = image_new(n, p);
    for (int i = 0; i < n; ++i) plane_release(p[i]);
    return s;
}
KC_CATCH

int kc_image_to_f32(kc_image *img, float *const host_planes[], int n)
try {
    KC_ARG(img && host_planes && n == img->n);
    Lock lk(ctx().mu);
    KC_TRY(image_force(img));  // one batched launch for R, G, B
    for (int i = 0; i < n; ++i) KC_TRY(kc_plane_download_f32(img->planes[i], host_planes[i], 0));
    return KC_OK;
}
KC_CATCH

int kc_image_read_png(const char *path, kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(path && out);
    std::vector<uint8_t> px;
    uint32_t w = 0, h = 0;
    int ch = 0;
    KC_TRY(png_read(path, px, w, h, ch));
    return kc_image_from_u8(px.data(), w, h, ch, out);
}
KC_CATCH

int kc_image_write_png(kc_image *img, const char *path)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(img && path);
    std::vector<uint8_t> px((size_t)img->w() * img->h() * 4);
    KC_TRY(image_to_u8(img, false, px.data()));
    return png_write_rgba8(path, px.data(), img->w(), img->h());
}
KC_CATCH

// ---------------------------------------------------------------- operators
int kc_calculate_size(int policy, const kc_size *sizes, int n, int slot_index, kc_size specific, kc_size *out)
try {
    KC_ARG(out && (n == 0 || sizes));
    return calculate_size(policy, sizes, n, slot_index, specific, out);
}
KC_CATCH

int kc_resize_image(kc_image *src, kc_size size, int filter, kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(src && out);
    return resize_image(src, size, filter, out);
}
KC_CATCH

int kc_resize_buffers(kc_image *const images[], const kc_edge keys[], int n, const kc_edge *edges_sorted, int n_edges,
                      int policy, uint32_t policy_slot, kc_size policy_size, int filter, kc_image *out[])
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(n >= 0 && (n == 0 || (images && keys && out)));
    if (n == 0) return KC_OK;  // shared.rs:147-149
    std::vector<kc_size> sizes;
    for (int i = 0; i < n; ++i) sizes.push_back(kc_size{ images[i]->w(), images[i]->h() });
    int slot_index = -1;
    if (policy == KC_POLICY_SPECIFIC_SLOT && edges_sorted && n_edges > 0) {
        const kc_edge *edge = nullptr;
        for (int i = 0; i < n_edges; ++i)
            if (edges_sorted[i].input_slot == policy_slot) {
                edge = &edges_sorted[i];
                break;
            }
        if (!edge) edge = &edges_sorted[0];
        for (int i = 0; i < n; ++i)
            if (keys[i].output_slot == edge->output_slot && keys[i].output_id == edge->output_id) {
                slot_index = i;
                break;
            }
    }
    kc_size size;
    KC_TRY(calculate_size(policy, sizes.data(), n, slot_index, policy_size, &size));
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    for (int i = 0; i < n; ++i) {
        if (images[i]->w() != size.width || images[i]->h() != size.height) {
            int s = resize_image(images[i], size, filter, &out[i]);
            if (s != KC_OK) {
                for (int j = 0; j < i; ++j) image_release(out[j]);
                return s;
            }
        } else {
            out[i] = images[i];
            image_retain(out[i]);
        }
    }
    return KC_OK;
}
KC_CATCH

int kc_mix_process(kc_image *left, kc_image *right, int mix_type, kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(out);
    return mix_process(left, right, mix_type, out);
}
KC_CATCH

int kc_separate_rgba_process(kc_image *input, kc_image *out[4])
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(out);
    return separate_process(input, out);
}
KC_CATCH

int kc_combine_rgba_process(kc_image *const inputs[4], kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(inputs && out);
    return combine_process(inputs, out);
}
KC_CATCH

int kc_value_process(float v, kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(out);
    return value_process(v, out);
}
KC_CATCH

int kc_height_to_normal_process(kc_image *input, kc_image **out)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    KC_ARG(out);
    return height_to_normal_process(input, out);
}
KC_CATCH

// ---------------------------------------------------------------- NodeGraph
int kc_node_graph_new(kc_node_graph **out)
try {
    KC_ARG(out);
    *out = new kc_node_graph();
    return KC_OK;
}
KC_CATCH

int kc_node_graph_clone(const kc_node_graph *g, kc_node_graph **out)
try {
    KC_ARG(g && out);
    *out = new kc_node_graph(*g);
    return KC_OK;
}
KC_CATCH

int kc_node_graph_free(kc_node_graph *g)
try {
    delete g;
    return KC_OK;
}
KC_CATCH

int kc_node_graph_from_json(const char *json, kc_node_graph **out)
try {
    KC_ARG(json && out);
    kc_node_graph *g = new kc_node_graph();
    int s = graph_from_json(json, g->g);
    if (s != KC_OK) {
        delete g;
        return s;
    }
    *out = g;
    return KC_OK;
}
KC_CATCH

int kc_node_graph_from_path(const char *path, kc_node_graph **out)
try {
    KC_ARG(path && out);
    std::ifstream f(path);
    if (!f) {
        set_error(std::string("cannot open ") + path);
        return KC_ERR_IO;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    return kc_node_graph_from_json(ss.str().c_str(), out);
}
KC_CATCH

int kc_node_graph_to_json(const kc_node_graph *g, char *buf, size_t cap, size_t *needed)
try {
    KC_ARG(g);
    std::string s = graph_to_json(g->g);
    if (needed) *needed = s.size() + 1;
    if (buf && cap > 0) {
        size_t n = std::min(cap - 1, s.size());
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return KC_OK;
}
KC_CATCH

int kc_node_graph_export_json(const kc_node_graph *g, const char *path)
try {
    KC_ARG(g && path);
    std::ofstream f(path);
    if (!f) {
        set_error(std::string("cannot create ") + path);
        return KC_ERR_IO;
    }
    f << graph_to_json(g->g);
    return f.good() ? KC_OK : KC_ERR_IO;
}
KC_CATCH

int kc_node_graph_add_node(kc_node_graph *g, const kc_node_desc *node, uint32_t *id)
try {
    KC_ARG(g && node);
    return g->g.add_node(node_from_desc(*node), id);
}
KC_CATCH

int kc_node_graph_add_node_with_id(kc_node_graph *g, const kc_node_desc *node)
try {
    KC_ARG(g && node);
    return g->g.add_node_with_id(node_from_desc(*node));
}
KC_CATCH

int kc_node_graph_connect(kc_node_graph *g, uint32_t on, uint32_t in, uint32_t os, uint32_t is)
try {
    KC_ARG(g);
    return g->g.connect(on, in, os, is);
}
KC_CATCH

int kc_node_graph_try_connect(kc_node_graph *g, uint32_t on, uint32_t in, uint32_t os, uint32_t is)
try {
    KC_ARG(g);
    return g->g.try_connect(on, in, os, is);
}
KC_CATCH

int kc_node_graph_remove_node(kc_node_graph *g, uint32_t id)
try {
    KC_ARG(g);
    return g->g.remove_node(id, nullptr);
}
KC_CATCH

int kc_node_graph_remove_edge(kc_node_graph *g, kc_edge e)
try {
    KC_ARG(g);
    return g->g.remove_edge(e);
}
KC_CATCH

int kc_node_graph_disconnect_slot(kc_node_graph *g, uint32_t id, int side, uint32_t slot)
try {
    KC_ARG(g);
    return g->g.disconnect_slot(id, side, slot, nullptr);
}
KC_CATCH

int kc_node_graph_node_count(const kc_node_graph *g, uint32_t *count)
try {
    KC_ARG(g && count);
    *count = (uint32_t)g->g.nodes.size();
    return KC_OK;
}
KC_CATCH

static int copy_ids(const std::vector<uint32_t> &v, uint32_t *ids, uint32_t cap, uint32_t *count)
{
    if (count) *count = (uint32_t)v.size();
    if (ids)
        for (uint32_t i = 0; i < cap && i < v.size(); ++i) ids[i] = v[i];
    return KC_OK;
}

static int copy_edges(const std::vector<kc_edge> &v, kc_edge *edges, uint32_t cap, uint32_t *count)
{
    if (count) *count = (uint32_t)v.size();
    if (edges)
        for (uint32_t i = 0; i < cap && i < v.size(); ++i) edges[i] = v[i];
    return KC_OK;
}

int kc_node_graph_node_ids(const kc_node_graph *g, uint32_t *ids, uint32_t cap, uint32_t *count)
try {
    KC_ARG(g);
    std::vector<uint32_t> v;
    for (auto &n : g->g.nodes) v.push_back(n.node_id);
    return copy_ids(v, ids, cap, count);
}
KC_CATCH

int kc_node_graph_edges(const kc_node_graph *g, kc_edge *edges, uint32_t cap, uint32_t *count)
try {
    KC_ARG(g);
    return copy_edges(g->g.edges, edges, cap, count);
}
KC_CATCH

int kc_node_graph_input_slot_id_with_name(const kc_node_graph *g, const char *name, uint32_t *slot)
try {
    KC_ARG(g && name && slot);
    for (auto &n : g->g.nodes)
        if (n.is_input() && n.text == name) {
            *slot = n.node_id;
            return KC_OK;
        }
    return KC_ERR_INVALID_NAME;
}
KC_CATCH

int kc_node_graph_output_slot_id_with_name(const kc_node_graph *g, const char *name, uint32_t *slot)
try {
    KC_ARG(g && name && slot);
    for (auto &n : g->g.nodes)
        if (n.is_output() && n.text == name) {
            *slot = n.node_id;
            return KC_OK;
        }
    return KC_ERR_INVALID_NAME;
}
KC_CATCH

int kc_node_graph_set_mix_type(kc_node_graph *g, uint32_t id, int mix)
try {
    KC_ARG(g && mix >= KC_MIX_ADD && mix <= KC_MIX_POW);
    Node *n = g->g.find(id);
    if (!n || n->type != KC_NODE_MIX) return KC_ERR_INVALID_NODE_ID;
    n->mix_type = mix;
    return KC_OK;
}
KC_CATCH

static int copy_name(const std::string &s, char *buf, size_t cap)
{
    if (buf && cap > 0) {
        size_t n = std::min(cap - 1, s.size());
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return KC_OK;
}

int kc_node_graph_set_image_node_path(kc_node_graph *g, uint32_t id, const char *path)
try {
    KC_ARG(g && path);
    Node *n = g->g.find(id);
    if (!n || n->type != KC_NODE_IMAGE) return KC_ERR_INVALID_NODE_ID;
    n->text = path;
    return KC_OK;
}
KC_CATCH

int kc_node_graph_rename_output_node(kc_node_graph *g, uint32_t id, const char *new_name, char *old_name, size_t cap)
try {
    KC_ARG(g && new_name);
    std::string old;
    KC_TRY(g->g.rename_output_node(id, new_name, &old));
    return copy_name(old, old_name, cap);
}
KC_CATCH

// ---------------------------------------------------------------- TextureProcessor / LiveGraph
int kc_tex_pro_new(uint64_t memory_threshold, kc_tex_pro **out)
try {
    KC_ARG(out);
    kc_tex_pro *tp = new kc_tex_pro();
    tp->memory_threshold = memory_threshold;
    *out = tp;
    return KC_OK;
}
KC_CATCH

int kc_tex_pro_free(kc_tex_pro *tp)
try {
    Lock api_lock(ctx().mu);  // reference counts are plain integers, see kc_plane::refs
    delete tp;
    return KC_OK;
}
KC_CATCH

int kc_tex_pro_new_live_graph(kc_tex_pro *tp, kc_live_graph **out)
try {
    KC_ARG(tp && out);
    kc_live_graph *lg = new kc_live_graph();
    lg->tp = tp;
    *out = lg;
    return KC_OK;
}
KC_CATCH

int kc_live_graph_free(kc_live_graph *lg)
try {
    if (!lg) return KC_OK;
    Lock lk(ctx().mu);
    delete lg;
    return KC_OK;
}
KC_CATCH

#define LG_LOCK(lg) KC_ARG(lg); Lock _ctx_lock(ctx().mu)

int kc_live_graph_set_flags(kc_live_graph *lg, int auto_update, int use_cache)
try {
    LG_LOCK(lg);
    lg->auto_update = auto_update != 0;
    lg->use_cache = use_cache != 0;
    return KC_OK;
}
KC_CATCH

int kc_live_graph_get_flags(const kc_live_graph *lg, int *auto_update, int *use_cache)
try {
    KC_ARG(lg);
    if (auto_update) *auto_update = lg->auto_update;
    if (use_cache) *use_cache = lg->use_cache;
    return KC_OK;
}
KC_CATCH

int kc_live_graph_set_node_graph(kc_live_graph *lg, const kc_node_graph *g)
try {
    LG_LOCK(lg);
    KC_ARG(g);
    lg->g = g->g;
    lg->reset_node_states();
    lg->clear_data();
    return KC_OK;
}
KC_CATCH

int kc_live_graph_node_graph(const kc_live_graph *lg, kc_node_graph **out)
try {
    KC_ARG(lg && out);
    kc_node_graph *g = new kc_node_graph();
    g->g = lg->g;
    *out = g;
    return KC_OK;
}
KC_CATCH

int kc_live_graph_add_node(kc_live_graph *lg, const kc_node_desc *node, uint32_t *id)
try {
    LG_LOCK(lg);
    KC_ARG(node);
    return lg->add_node(node_from_desc(*node), id);
}
KC_CATCH

int kc_live_graph_add_node_with_id(kc_live_graph *lg, const kc_node_desc *node)
try {
    LG_LOCK(lg);
    KC_ARG(node);
    return lg->add_node_with_id(node_from_desc(*node));
}
KC_CATCH

int kc_live_graph_remove_node(kc_live_graph *lg, uint32_t id)
try {
    LG_LOCK(lg);
    return lg->remove_node(id);
}
KC_CATCH

int kc_live_graph_connect(kc_live_graph *lg, uint32_t on, uint32_t in, uint32_t os, uint32_t is)
try {
    LG_LOCK(lg);
    return lg->connect(on, in, os, is);
}
KC_CATCH

int kc_live_graph_remove_edge(kc_live_graph *lg, kc_edge e)
try {
    LG_LOCK(lg);
    return lg->remove_edge(e);
}
KC_CATCH

int kc_live_graph_disconnect_slot(kc_live_graph *lg, uint32_t id, int side, uint32_t slot)
try {
    LG_LOCK(lg);
    return lg->disconnect_slot(id, side, slot);
}
KC_CATCH

int kc_live_graph_set_mix_type(kc_live_graph *lg, uint32_t id, int mix)
try {
    LG_LOCK(lg);
    KC_ARG(mix >= KC_MIX_ADD && mix <= KC_MIX_POW);
    Node *n = lg->g.find(id);
    if (!n) return KC_ERR_INVALID_NODE_ID;
    KC_TRY(lg->set_state(id, KC_STATE_DIRTY));  // node_mut, :369-374
    if (n->type != KC_NODE_MIX) return KC_ERR_INVALID_NODE_TYPE;
    n->mix_type = mix;
    return KC_OK;
}
KC_CATCH

int kc_live_graph_rename_output_node(kc_live_graph *lg, uint32_t id, const char *new_name, char *old_name, size_t cap)
try {
    LG_LOCK(lg);
    KC_ARG(new_name);
    std::string old;
    KC_TRY(lg->g.rename_output_node(id, new_name, &old));
    return copy_name(old, old_name, cap);
}
KC_CATCH

int kc_live_graph_set_resize(kc_live_graph *lg, uint32_t id, int policy, uint32_t slot, kc_size size, int filter)
try {
    LG_LOCK(lg);
    KC_ARG(policy >= 0 && policy <= KC_POLICY_SPECIFIC_SIZE && filter >= 0 && filter <= KC_FILTER_LANCZOS3);
    Node *n = lg->g.find(id);
    if (!n) return KC_ERR_INVALID_NODE_ID;
    KC_TRY(lg->set_state(id, KC_STATE_DIRTY));
    n->policy = policy;
    n->policy_slot = slot;
    n->policy_size = size;
    n->filter = filter;
    return KC_OK;
}
KC_CATCH

int kc_live_graph_node_state(const kc_live_graph *lg, uint32_t id, int *state)
try {
    KC_ARG(lg && state);
    return lg->state_of(id, state);
}
KC_CATCH

int kc_live_graph_request(kc_live_graph *lg, uint32_t id)
try {
    LG_LOCK(lg);
    int st;
    KC_TRY(lg->state_of(id, &st));
    if (st == KC_STATE_DIRTY) lg->node_state[id] = KC_STATE_REQUESTED;
    return KC_OK;
}
KC_CATCH

int kc_live_graph_prioritise(kc_live_graph *lg, uint32_t id)
try {
    LG_LOCK(lg);
    int st;
    KC_TRY(lg->state_of(id, &st));
    if (st == KC_STATE_DIRTY || st == KC_STATE_REQUESTED) lg->node_state[id] = KC_STATE_PRIORITISED;
    return KC_OK;
}
KC_CATCH

int kc_live_graph_await_clean(kc_live_graph *lg, uint32_t id)
try {
    LG_LOCK(lg);
    return lg->await_clean(id);
}
KC_CATCH

int kc_live_graph_update(kc_live_graph *lg)
try {
    LG_LOCK(lg);
    return lg->update();
}
KC_CATCH

int kc_live_graph_slot_data(kc_live_graph *lg, uint32_t node, uint32_t slot, kc_image **out)
try {
    LG_LOCK(lg);
    KC_ARG(out);
    const SlotData *sd = lg->find_slot(node, slot);
    if (!sd) return KC_ERR_NO_SLOT_DATA;
    image_retain(sd->image);
    *out = sd->image;
    return KC_OK;
}
KC_CATCH

int kc_live_graph_slot_data_size(kc_live_graph *lg, uint32_t node, uint32_t slot, kc_size *size)
try {
    LG_LOCK(lg);
    KC_ARG(size);
    const SlotData *sd = lg->find_slot(node, slot);
    if (!sd) return KC_ERR_NO_SLOT_DATA;
    *size = kc_size{ sd->image->w(), sd->image->h() };
    return KC_OK;
}
KC_CATCH

int kc_live_graph_slot_in_memory(kc_live_graph *lg, uint32_t node, uint32_t slot, int *in_memory)
try {
    LG_LOCK(lg);
    KC_ARG(in_memory);
    if (!lg->find_slot(node, slot)) return KC_ERR_NO_SLOT_DATA;
    *in_memory = 1;  // planes never leave HBM: there is no disk tier to page out to
    return KC_OK;
}
KC_CATCH

int kc_live_graph_node_slot_ids(kc_live_graph *lg, uint32_t node, uint32_t *slots, uint32_t cap, uint32_t *count)
try {
    LG_LOCK(lg);
    std::vector<uint32_t> v;
    for (auto &sd : lg->slots_of(node)) v.push_back(sd.slot_id);
    return copy_ids(v, slots, cap, count);
}
KC_CATCH

int kc_live_graph_buffer_rgba(kc_live_graph *lg, uint32_t node, uint32_t slot, int srgb, uint8_t *host)
try {
    LG_LOCK(lg);
    KC_ARG(host);
    const SlotData *sd = lg->find_slot(node, slot);
    if (!sd) return KC_ERR_NO_SLOT_DATA;
    return image_to_u8(sd->image, srgb != 0, host);
}
KC_CATCH

int kc_live_graph_embed_slot_data_with_id(kc_live_graph *lg, kc_image *image, uint32_t slot_id, uint32_t embed_id)
try {
    LG_LOCK(lg);
    KC_ARG(image);
    for (auto &e : lg->embedded)
        if (e.slot_data_id == embed_id) return KC_ERR_INVALID_SLOT_ID;  // :329-340
    image_retain(image);
    lg->embedded.push_back(EmbeddedSlotData{ embed_id, slot_id, image });
    return KC_OK;
}
KC_CATCH

// ---------------------------------------------------------------- row bands (bands.cpp)
int kc_live_graph_embed_slot_data_band(kc_live_graph *lg, kc_image *image, uint32_t slot_id, uint32_t embed_id, int32_t band_y0,
                                       uint32_t full_height)
try {
    LG_LOCK(lg);
    KC_ARG(image && full_height > 0);
    for (auto &e : lg->embedded)
        if (e.slot_data_id == embed_id) return KC_ERR_INVALID_SLOT_ID;
    image_retain(image);
    EmbeddedSlotData e{ embed_id, slot_id, image };
    e.band_y0 = band_y0;
    e.full_h = full_height;
    lg->embedded.push_back(e);
    return KC_OK;
}
KC_CATCH

int kc_live_graph_evaluate_band(kc_live_graph *lg, uint32_t node_id, uint32_t slot_id, int32_t y0, int32_t y1, kc_image **out)
try {
    LG_LOCK(lg);
    KC_ARG(out);
    return band_evaluate(*lg, node_id, slot_id, y0, y1, out);
}
KC_CATCH

int kc_live_graph_band_source_rows(kc_live_graph *lg, uint32_t node_id, int32_t y0, int32_t y1, kc_band_rows *rows, uint32_t cap,
                                   uint32_t *count)
try {
    LG_LOCK(lg);
    KC_ARG(count);
    std::vector<kc_band_rows> v;
    KC_TRY(band_source_rows(*lg, node_id, y0, y1, v));
    *count = (uint32_t)v.size();
    for (uint32_t i = 0; rows && i < cap && i < *count; ++i) rows[i] = v[i];
    return KC_OK;
}
KC_CATCH

int kc_live_graph_add_input_slot_data(kc_live_graph *lg, uint32_t node_id, uint32_t slot_id, kc_image *image)
try {
    LG_LOCK(lg);
    KC_ARG(image);
    image_retain(image);
    lg->input_slot_datas.push_back(SlotData{ node_id, slot_id, image });
    return KC_OK;
}
KC_CATCH

// ---------------------------------------------------------------- the u8 boundary as a pipeline (u8pipe.cpp)
int kc_u8_pipe_create(uint32_t width, uint32_t height, int channels, int depth, kc_u8_pipe **out)
try {
    Lock lk(ctx().mu);
    KC_ARG(out);
    return u8_pipe_create(width, height, channels, depth, out);
}
KC_CATCH

int kc_u8_pipe_free(kc_u8_pipe *pipe)
try {
    Lock lk(ctx().mu);
    return u8_pipe_free(pipe);
}
KC_CATCH

int kc_u8_pipe_buffers(kc_u8_pipe *pipe, int slot, uint8_t **host_in, const uint8_t **host_out)
try {
    Lock lk(ctx().mu);
    KC_ARG(pipe);
    return u8_pipe_buffers(pipe, slot, host_in, host_out);
}
KC_CATCH

int kc_u8_pipe_upload(kc_u8_pipe *pipe, int slot, kc_image **out)
try {
    Lock lk(ctx().mu);
    KC_ARG(pipe && out);
    return u8_pipe_upload(pipe, slot, out);
}
KC_CATCH

int kc_u8_pipe_download(kc_u8_pipe *pipe, int slot, kc_image *img, int srgb)
try {
    Lock lk(ctx().mu);
    KC_ARG(pipe && img);
    return u8_pipe_download(pipe, slot, img, srgb != 0);
}
KC_CATCH

int kc_u8_pipe_wait_download(kc_u8_pipe *pipe, int slot)
try {
    KC_ARG(pipe);
    kc_u8_pipe *p = pipe;
    (void)p;
    return u8_pipe_wait_download(pipe, slot);  // no context lock: only this call blocks, and only on the slot's own event
}
KC_CATCH

// ---------------------------------------------------------------- multi-GPU placement (partition.cpp)
int kc_live_graph_partition(kc_live_graph *lg, uint32_t root, int world, int policy, kc_partition **out)
try {
    LG_LOCK(lg);
    KC_ARG(out);
    return partition_plan(*lg, root, world, policy, out);
}
KC_CATCH

int kc_partition_free(kc_partition *p)
{
    delete p;
    return KC_OK;
}

int kc_partition_info(const kc_partition *p, int *world, int *home, int *levels)
try {
    KC_ARG(p);
    if (world) *world = p->world;
    if (home) *home = p->home;
    if (levels) *levels = p->n_levels;
    return KC_OK;
}
KC_CATCH

int kc_partition_nodes(const kc_partition *p, kc_placement *out, uint32_t cap, uint32_t *count)
try {
    KC_ARG(p && count);
    *count = (uint32_t)p->nodes.size();
    for (uint32_t i = 0; out && i < cap && i < *count; ++i) out[i] = p->nodes[i];
    return KC_OK;
}
KC_CATCH

int kc_partition_transfers(const kc_partition *p, kc_transfer *out, uint32_t cap, uint32_t *count)
try {
    KC_ARG(p && count);
    *count = (uint32_t)p->xfers.size();
    for (uint32_t i = 0; out && i < cap && i < *count; ++i) out[i] = p->xfers[i];
    return KC_OK;
}
KC_CATCH

int kc_partition_kind(const kc_partition *p, int *kind, double *est_single, double *est_branches, double *est_bands)
try {
    KC_ARG(p);
    if (kind) *kind = p->kind;
    if (est_single) *est_single = p->est_single;
    if (est_branches) *est_branches = p->est_branches;
    if (est_bands) *est_bands = p->est_bands;
    return KC_OK;
}
KC_CATCH

int kc_partition_bands(const kc_partition *p, kc_band_range *out, uint32_t cap, uint32_t *count, uint32_t *full_width, uint32_t *full_height)
try {
    KC_ARG(p && count);
    *count = (uint32_t)p->bands.size();
    for (uint32_t i = 0; out && i < cap && i < *count; ++i) out[i] = p->bands[i];
    if (full_width) *full_width = p->full_w;
    if (full_height) *full_height = p->full_h;
    return KC_OK;
}
KC_CATCH

int kc_partition_set_gather(kc_partition *p, int gather)
try {
    KC_ARG(p);
    p->gather = gather != 0;
    return KC_OK;
}
KC_CATCH

int kc_comm_unique_id(void *id)
try {
    return comm_unique_id(id, KC_COMM_ID_BYTES);
}
KC_CATCH

int kc_comm_init(int rank, int world_size, const void *id)
try {
    Lock lk(ctx().mu);
    return comm_init(rank, world_size, id, KC_COMM_ID_BYTES);
}
KC_CATCH

int kc_comm_destroy(void)
try {
    Lock lk(ctx().mu);
    return comm_destroy();
}
KC_CATCH

int kc_comm_info(int *rank, int *world_size)
try {
    Lock lk(ctx().mu);
    comm_info(rank, world_size);
    return KC_OK;
}
KC_CATCH

int kc_comm_stats(uint64_t *planes_sent, uint64_t *planes_received, uint64_t *bytes_sent)
try {
    Lock lk(ctx().mu);
    comm_stats(planes_sent, planes_received, bytes_sent);
    return KC_OK;
}
KC_CATCH

int kc_comm_transport(char *buf, size_t cap)
try {
    Lock lk(ctx().mu);
    KC_ARG(buf && cap > 0);
    std::snprintf(buf, cap, "%s", comm_wire_name());
    return KC_OK;
}
KC_CATCH

int kc_comm_gather_bands(kc_image *band, int32_t y0, uint32_t full_height, int home_rank, kc_image **out)
try {
    Lock lk(ctx().mu);
    KC_ARG(band && out);
    return comm_gather_bands(band, y0, full_height, home_rank, out);
}
KC_CATCH

int kc_live_graph_exchange(kc_live_graph *lg, const kc_transfer *transfers, uint32_t count)
try {
    LG_LOCK(lg);
    KC_ARG(transfers || count == 0);
    return comm_exchange(*lg, transfers, count);
}
KC_CATCH

int kc_live_graph_evaluate_partitioned(kc_live_graph *lg, const kc_partition *plan, uint32_t root_node_id, kc_image **out)
try {
    LG_LOCK(lg);
    KC_ARG(plan);
    return comm_evaluate_partitioned(*lg, *plan, root_node_id, out);
}
KC_CATCH

int kc_live_graph_import_slot_data(kc_live_graph *lg, uint32_t node_id, uint32_t slot_id, kc_image *image)
try {
    LG_LOCK(lg);
    KC_ARG(image);
    return lg->import_slot_data(node_id, slot_id, image);
}
KC_CATCH

int kc_live_graph_changed_consume(kc_live_graph *lg, uint32_t *ids, uint32_t cap, uint32_t *count)
try {
    LG_LOCK(lg);
    std::vector<uint32_t> v(lg->changed.begin(), lg->changed.end());
    std::sort(v.begin(), v.end());
    if (ids && cap >= v.size()) lg->changed.clear();  // a NULL / short buffer only queries the count
    return copy_ids(v, ids, cap, count);
}
KC_CATCH

int kc_live_graph_output_ids(const kc_live_graph *lg, uint32_t *ids, uint32_t cap, uint32_t *count)
try {
    KC_ARG(lg);
    return copy_ids(lg->g.output_ids(), ids, cap, count);
}
KC_CATCH

int kc_live_graph_node_ids(const kc_live_graph *lg, uint32_t *ids, uint32_t cap, uint32_t *count)
try {
    KC_ARG(lg);
    std::vector<uint32_t> v;
    for (auto &n : lg->g.nodes) v.push_back(n.node_id);
    return copy_ids(v, ids, cap, count);
}
KC_CATCH

int kc_live_graph_edges(const kc_live_graph *lg, kc_edge *edges, uint32_t cap, uint32_t *count)
try {
    KC_ARG(lg);
    return copy_edges(lg->g.edges, edges, cap, count);
}
KC_CATCH

int kc_live_graph_set_base_dir(kc_live_graph *lg, const char *dir)
try {
    LG_LOCK(lg);
    lg->base_dir = dir ? dir : "";
    return KC_OK;
}
KC_CATCH

}  // extern "C"
