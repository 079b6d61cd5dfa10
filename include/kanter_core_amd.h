/*
 * kanter_core_amd.h -- C ABI of the MI355X (gfx950) per-pixel evaluation backend for
 * kanter_core / vismut_core 0.10.0 graphs.
 *
 * This is the drop-in boundary: plain pointers, sizes and opaque handles, no C++ or torch types.
 * The reference (pure Rust, no FFI of its own) would bind these from an `extern "C"` block and
 * call them from the bodies behind its operator boundary, `process_node`
 * (src/node/node_type.rs:213-248) and the per-node `process` functions it dispatches to
 * (src/node/node_type.rs:107-122).  Each entry point cites the reference interface it replaces;
 * paths are relative to the reference checkout.  INTEGRATION.md shows the Rust-side binding.
 *
 * Data model (replaces src/slot_image.rs:12-19 and src/transient_buffer.rs):
 *   kc_plane  one channel: row-major f32, `pitch` bytes per row (pitch >= 4*width, 256-byte
 *             aligned for planes this library allocates) living in HBM -- or a broadcast
 *             constant / a not-yet-materialised pointwise chain (see kc_plane_materialize).
 *             Planes are immutable once published and shared by reference count exactly where
 *             the reference clones an Arc<TransientBufferContainer>.
 *   kc_image  SlotImage: Gray = 1 plane, Rgba = 4 planes (R, G, B, A).
 *
 * All functions return a kc_status (0 = ok).  1..19 mirror TexProError (src/error.rs:5-27);
 * >= 100 are backend errors.  Every entry point is thread-safe; work is enqueued in order on
 * one HIP stream per process (kc_set_stream) and is complete after kc_sync() or any call that
 * returns host data.
 */
#ifndef KANTER_CORE_AMD_H
#define KANTER_CORE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KC_API __attribute__((visibility("default")))

/* ---- status codes: TexProError, src/error.rs:5-27 --------------------------------------- */
typedef enum kc_status {
    KC_OK = 0,
    KC_ERR_GENERIC = 1,
    KC_ERR_CANCELED = 2,
    KC_ERR_IMAGE = 3,
    KC_ERR_INVALID_BUFFER_COUNT = 4,
    KC_ERR_INVALID_NODE_ID = 5,
    KC_ERR_INVALID_NODE_TYPE = 6,
    KC_ERR_INVALID_SLOT_ID = 7,
    KC_ERR_INVALID_SLOT_TYPE = 8,
    KC_ERR_INVALID_EDGE = 9,
    KC_ERR_NO_SLOT_DATA = 10,
    KC_ERR_SLOT_OCCUPIED = 11,
    KC_ERR_SLOT_NOT_OCCUPIED = 12,
    KC_ERR_UNABLE_TO_LOCK = 13,
    KC_ERR_NODE_PROCESSING = 14,
    KC_ERR_POISON = 15,
    KC_ERR_TRY_LOCK = 16,
    KC_ERR_NODE_DIRTY = 17,
    KC_ERR_IO = 18,
    KC_ERR_INVALID_NAME = 19,
    /* backend */
    KC_ERR_HIP = 100,          /* a HIP runtime call failed; kc_last_error() has the text */
    KC_ERR_NO_DEVICE = 101,    /* no gfx950 device / kc_init not called: there is NO CPU fallback */
    KC_ERR_INVALID_ARG = 102,
    KC_ERR_OUT_OF_MEMORY = 103,
    KC_ERR_UNSUPPORTED = 104
} kc_status;

/* MixType, src/node/mix.rs:20-27 */
typedef enum kc_mix_type { KC_MIX_ADD = 0, KC_MIX_SUBTRACT = 1, KC_MIX_MULTIPLY = 2, KC_MIX_DIVIDE = 3, KC_MIX_POW = 4 } kc_mix_type;
/* ResizeFilter, src/node/mod.rs:62-69 (default Triangle, :71-75) */
typedef enum kc_resize_filter { KC_FILTER_NEAREST = 0, KC_FILTER_TRIANGLE = 1, KC_FILTER_CATMULLROM = 2, KC_FILTER_GAUSSIAN = 3, KC_FILTER_LANCZOS3 = 4 } kc_resize_filter;
/* ResizePolicy, src/node/mod.rs:33-41 (default MostPixels, :43-47) */
typedef enum kc_resize_policy { KC_POLICY_MOST_PIXELS = 0, KC_POLICY_LEAST_PIXELS = 1, KC_POLICY_LARGEST_AXES = 2, KC_POLICY_SMALLEST_AXES = 3, KC_POLICY_SPECIFIC_SLOT = 4, KC_POLICY_SPECIFIC_SIZE = 5 } kc_resize_policy;
/* NodeType, src/node/node_type.rs:14-28 */
typedef enum kc_node_type {
    KC_NODE_INPUT_GRAY = 0, KC_NODE_INPUT_RGBA = 1, KC_NODE_OUTPUT_GRAY = 2, KC_NODE_OUTPUT_RGBA = 3,
    KC_NODE_GRAPH = 4, KC_NODE_IMAGE = 5, KC_NODE_EMBED = 6, KC_NODE_WRITE = 7, KC_NODE_VALUE = 8,
    KC_NODE_MIX = 9, KC_NODE_HEIGHT_TO_NORMAL = 10, KC_NODE_SEPARATE_RGBA = 11, KC_NODE_COMBINE_RGBA = 12
} kc_node_type;
/* NodeState, src/live_graph.rs:22-37 */
typedef enum kc_node_state { KC_STATE_CLEAN = 0, KC_STATE_DIRTY = 1, KC_STATE_REQUESTED = 2, KC_STATE_PRIORITISED = 3, KC_STATE_PROCESSING = 4, KC_STATE_PROCESSING_DIRTY = 5 } kc_node_state;
/* Side, src/node/mod.rs:101-105 */
typedef enum kc_side { KC_SIDE_INPUT = 0, KC_SIDE_OUTPUT = 1 } kc_side;

typedef struct kc_plane kc_plane;
typedef struct kc_image kc_image;
typedef struct kc_node_graph kc_node_graph;
typedef struct kc_live_graph kc_live_graph;
typedef struct kc_tex_pro kc_tex_pro;
typedef struct kc_partition kc_partition;   /* multi-GPU placement plan of one graph evaluation */
typedef struct kc_u8_pipe kc_u8_pipe;       /* pipelined u8 host boundary (pinned buffers, copy streams) */

/* Size, src/slot_data.rs:4-30 */
typedef struct kc_size { uint32_t width, height; } kc_size;
/* Edge, src/edge.rs:8-14 */
typedef struct kc_edge { uint32_t output_id, input_id, output_slot, input_slot; } kc_edge;

/* Node, src/node/mod.rs:113-123 (priority / cancel are host scheduling state, not carried). */
typedef struct kc_node_desc {
    uint32_t node_id;          /* NodeId */
    int32_t node_type;         /* kc_node_type */
    int32_t mix_type;          /* kc_mix_type, for KC_NODE_MIX */
    float value;               /* for KC_NODE_VALUE */
    uint32_t embed_id;         /* EmbeddedSlotDataId, for KC_NODE_EMBED */
    const char *text;          /* Input/Output name, Image/Write path; may be NULL */
    const kc_node_graph *graph;/* nested graph for KC_NODE_GRAPH (copied) */
    int32_t resize_policy;     /* kc_resize_policy */
    uint32_t policy_slot;      /* SpecificSlot(SlotId) */
    kc_size policy_size;       /* SpecificSize(Size) */
    int32_t resize_filter;     /* kc_resize_filter */
} kc_node_desc;

/* ========================================================================================== *
 * Device / context
 * ========================================================================================== */
/* Binds the process to one gfx950 device (one process per GPU).  Fails with KC_ERR_NO_DEVICE
 * when no GPU is present -- the library never computes on the CPU. */
KC_API int kc_init(int device_ordinal);
KC_API int kc_shutdown(void);
KC_API int kc_is_initialized(void);
/* Enqueue all work on the caller's hipStream_t (e.g. torch's current stream); NULL restores the
 * library's own stream.  Replaces the reference's thread-per-node scheduling, src/engine.rs:288. */
KC_API int kc_set_stream(void *hip_stream);
KC_API void *kc_get_stream(void);
KC_API int kc_sync(void);
KC_API const char *kc_last_error(void);
KC_API const char *kc_status_string(int status);
/* 1 (default): pointwise Mix chains whose intermediates are never observed are evaluated by one
 * fused kernel; 0: every node materialises its planes.  Results are bit-identical either way. */
KC_API int kc_set_fusion(int enabled);
KC_API int kc_get_fusion(void);
/* Which resize kernels may run (an A/B and test knob; results are bit-identical in every mode; env KC_RESIZE_MODE):
 * 0 (default) all; 1 no resize_poly_kernel; 2 no resize_down_kernel either; 3 two passes through HBM only;
 * 4 everything except the integer-ratio up-sampling kernels.  Replaces nothing of the reference
 * (src/shared.rs:159-199 has one code path). */
KC_API int kc_set_resize_mode(int mode);
KC_API int kc_get_resize_mode(void);
/* Cache policy (a throughput knob; results are identical): 1 (default, env KC_CACHE_POLICY) = a launch that streams more
 * than the 256 MB Infinity Cache can hold reads its full-size inputs with the nontemporal hint and keeps its result
 * cacheable while that fits; 0 = plain loads and stores everywhere. */
KC_API int kc_set_cache_policy(int mode);
KC_API int kc_get_cache_policy(void);
/* Named A/B and test switches (results are identical whatever they say; unknown names are refused):
 *   "chain1" 1 (default): a single Mix step runs its ahead-of-time straight-line kernel; 0: the step interpreter / the
 *   run-time specialiser, as longer programs do;
 *   "replay" 1 (default): an evaluation that repeats the previous one of the same node exactly (same graph by content, same
 *   node states, same slot data and embedded images by identity) skips the node-by-node walk of src/engine.rs:200-307 and
 *   re-issues the recorded launches; 0: always walk.
 *   "join" 1 (default): a Mix whose two inputs are both chains that have not run keeps both in one program -- one launch,
 *   no plane in between (csrc/runtime.cpp plane_mix; only kernels compiled at run time implement it, and while such a kernel
 *   is not there the second chain runs on its own as before); 0: always run it on the spot.  Bit-identical either way.
 *   "wide" 1 (default): a fused chain may read up to 16 planes per channel (kernels compiled at run time only; cut to the
 *   interpreter's 4 while such a kernel is not there); 0: 4, as before.  Bit-identical either way.
 *   "down2" 0 / 1 (default) / 2: down-sampling with more than 8 taps on both axes runs resize_down2_kernel never / except
 *   where the integer-ratio streaming kernel runs at ratio 4 or 8 / wherever its tables exist (bit-identical; A/B and tests).
 *   "down2_by_rows" -1 (default) / 0 / 1: resize_down2_kernel's job order -- four strips of one row group per workgroup and the
 *   XCDs' eighths row by row (1), four row groups of one strip (0), or by the table (-1: 1 where the windows span several chunks).
 *   "poly2" 1 (default) / 0, "poly2_min_ratio" 8 (default; 2, 4): down-sampling with an integer vertical ratio of at least
 *   poly2_min_ratio and windows of 4 or 6 ages runs resize_poly2_kernel (two waves per band strip) / the kernels it replaces
 *   (bit-identical; A/B and tests).
 *   "link_gbps" (153), "hbm_gbps" (6100): the rates kc_live_graph_partition prices a transfer / a streaming kernel with.
 *   "cache_budget_mb" (208, env KC_CACHE_BUDGET_MB): how much of a launch's streams the cache policy leaves cacheable -- 13/16 of
 *   the MI355X's 256 MB Infinity Cache; HIP reports no size for that cache, so another part sets this. */
KC_API int kc_set_option(const char *name, int value);
KC_API int kc_get_option(const char *name, int *value);
/* Diagnostics (host only, works without a device): the structure the integer-ratio up-sampling kernels rely on,
 * for one axis of image::imageops::resize (src/shared.rs:159-199) from in_n to out_n samples with `filter`.
 * *eligible = 0: the tap table does not have it (not a whole ratio, an even window ...) and the general kernels run.
 * Otherwise info = { ratio, taps, off, b_lo, b_hi }: the window of output o is source samples
 * [o / ratio - off, o / ratio - off + taps) cut to the source; outputs b_lo .. out_n - b_hi - 1 use weight row
 * o % ratio, the first b_lo and last b_hi outputs rows ratio + o and ratio + b_lo + (o - (out_n - b_hi));
 * `rows` receives those (ratio + b_lo + b_hi) x taps weights (as many as fit `cap` floats). */
KC_API int kc_resize_upsample_plan(uint32_t in_n, uint32_t out_n, int filter, int *eligible, int32_t info[5], float *rows,
                                   size_t cap);
/* Diagnostics (host only): what resize_down2_kernel (csrc/down2.hip: down-sampling with more than 8 taps on both axes) reads for
 * one axis of image::imageops::resize (src/shared.rs:159-199).  info = { stride (most taps of any output), nc, hstride, tile_w,
 * fewest taps of any output }.  nc = records per group of four outputs when the table serves as the VERTICAL one (0: it cannot --
 * at most 8 taps, or some group's windows span more than 64 source samples); hstride / tile_w = row pitch of the padded weights
 * and strip width when it serves as the HORIZONTAL one (0: it cannot).  Optional outputs: left_count = out_n window starts then
 * out_n tap counts, w = out_n x stride weights of the plain table, vrec = ceil(out_n / 4) x nc records of 72 dwords ([0] first
 * source sample, [1], [2] presence mask of tap (sample u, output k) at bit 4 u + k, [3] last sample of the group's windows, [4]
 * records in use, [5] the same in halves of 8 samples, [8 + 4 u + k] weights), hw = out_n x hstride padded weights; each as many elements as its capacity allows. */
KC_API int kc_resize_down2_plan(uint32_t in_n, uint32_t out_n, int filter, int32_t info[5], uint32_t *left_count, float *w, size_t wcap,
                                uint32_t *vrec, size_t vcap, float *hw, size_t hcap);
/* Pool statistics: bytes currently handed out, bytes cached for reuse, kernels launched. */
KC_API int kc_stats(uint64_t *bytes_in_use, uint64_t *bytes_cached, uint64_t *kernel_launches);
/* Algorithmic HBM bytes of every kernel launched so far: per launch, each resident input plane read once and each
 * result plane written once (what a roofline divides by; fused intermediates and constant planes cost nothing). */
KC_API int kc_stats_algorithmic_bytes(uint64_t *bytes);
/* Named event counters since kc_init (tests and profiling: which kernel family a call went through).  Unknown names
 * read 0.  Names: "upsample_launches", "upsample_chain_launches" (the integer-ratio up-sampling kernels),
 * "resize_chain_launches" (the general fused resample + chain kernel), "chain1_launches" (one-step programs through the
 * ahead-of-time kernels), "replayed_evaluations". */
KC_API int kc_stats_counter(const char *name, uint64_t *value);
KC_API int kc_pool_trim(void);
/* Run-time specialisation of the fused Mix-chain kernel.  A chain of N Mix nodes (src/node/mix.rs:136-192
 * applied N times) normally runs through a step-table interpreter; a program that keeps coming back is also
 * emitted as straight-line HIP, compiled with hiprtc (same parity flags as the offline build) and used from
 * then on.  Results are bit-identical either way; this is a throughput knob only.
 *   mode 0: interpreter only.  mode 1 (default, env KC_SPECIALIZE): compile in the background once a program
 *   has been seen `after` times (default 2; <= 0 keeps the current value); launches never wait.
 *   mode 2: compile at the first sighting, the launch waits for the compiler (tests, batch jobs).
 * kc_specialize_wait blocks until every queued compile has landed. */
KC_API int kc_set_specialize(int mode, int after);
KC_API int kc_get_specialize(void);
KC_API int kc_specialize_wait(void);
KC_API int kc_specialize_stats(uint64_t *kernels_compiled, uint64_t *compiles_failed, uint64_t *specialized_launches,
                               uint64_t *compiles_pending);
/* Diagnostics: generates the specialised kernel for a program given as step words (ChainCode | (source + 1) << 8,
 * source = -1 for the step's constant, k for input plane k) and compiles it for gfx950 WITHOUT loading it -- works
 * without a device.  The generated source is copied to `source` (NUL-terminated, truncated to `cap`) when given. */
KC_API int kc_specialize_compile_check(const uint32_t *words, uint32_t n_ops, uint32_t n_in, int start_src, int flat,
                                       char *source, size_t cap);
/* The same for a program that runs inside the integer-ratio up-sampling kernel (the resampled operand is input slot
 * n_in - 1; `taps` = 1 or 3 per axis, `wide` = the 1024-column tile form): src/shared.rs:159-199 feeding
 * src/node/mix.rs:136-192 in one launch. */
KC_API int kc_specialize_compile_check_upsample(const uint32_t *words, uint32_t n_ops, uint32_t n_in, int start_src, uint32_t taps,
                                                int wide, char *source, size_t cap);
/* Compiled kernels outlive the process: every code object hiprtc produces is written to a directory, and the FIRST sighting of a
 * program in a later process loads it from there (no compile, no interpreter run) -- the reference's own usage is one evaluation
 * per process (tests/integration_tests.rs:47-49).  Directory: KC_KERNEL_CACHE_DIR, default $XDG_CACHE_HOME/kanter_core_amd or
 * ~/.cache/kanter_core_amd ("off": none); read-only second place: kernel_cache/ next to the library, filled by the build with
 * the BASELINE programs.  A file is trusted only if its signature equals the program's byte for byte, the hash of what TODAY's
 * generator, options and hiprtc version produce for it equals the stored one and the code's checksum holds; anything else is
 * ignored and replaced by a fresh compile.
 *   kc_kernel_cache_set_dir   NULL / "": the environment's choice again; "off": no cache;
 *   kc_kernel_cache_stats     files accepted / refused / written, kernels this process took from files;
 *   kc_kernel_cache_precompile  compiles the program given by value (step words as for kc_specialize_compile_check, cache-policy
 *                             mask, up_taps > 0: the up-sampling form) WITHOUT a device and writes its file into `dir`;
 *   kc_specialize_reset       forgets the kernels this process holds (files stay): the next sighting is a first one. */
KC_API int kc_kernel_cache_set_dir(const char *dir);
KC_API int kc_kernel_cache_stats(uint64_t *files_accepted, uint64_t *files_refused, uint64_t *files_written, uint64_t *kernels_loaded);
KC_API int kc_kernel_cache_precompile(const uint32_t *words, uint32_t n_ops, uint32_t n_in, int start_src, int flat, uint32_t nt_mask,
                                      uint32_t up_taps, int up_wide, const char *dir);
KC_API int kc_specialize_reset(void);

/* ========================================================================================== *
 * Planes -- replaces Buffer / TransientBufferContainer (src/slot_image.rs:12,
 * src/transient_buffer.rs:188-247); HBM replaces the RAM/disk tiering.
 * ========================================================================================== */
KC_API int kc_plane_alloc(uint32_t width, uint32_t height, kc_plane **out);
/* Broadcast constant: what `vec![v; n]` (src/slot_image.rs:28-64) and the 1x1 Value plane
 * (src/node/mod.rs:240-244) hold, kept as a scalar until somebody needs the bytes. */
KC_API int kc_plane_const(uint32_t width, uint32_t height, float value, kc_plane **out);
/* Wrap caller-owned device memory (e.g. a torch tensor); not freed by the library. */
KC_API int kc_plane_wrap(void *device_ptr, uint32_t width, uint32_t height, size_t pitch_bytes, kc_plane **out);
KC_API int kc_plane_retain(kc_plane *p);
KC_API int kc_plane_release(kc_plane *p);
KC_API int kc_plane_size(const kc_plane *p, uint32_t *width, uint32_t *height);
KC_API int kc_plane_is_const(const kc_plane *p, int *is_const, float *value);
/* Forces the plane into HBM (runs any pending fused chain, fills constants). */
KC_API int kc_plane_materialize(kc_plane *p);
/* Materialises, then returns the device pointer and pitch (valid while the plane is retained). */
KC_API int kc_plane_device_ptr(kc_plane *p, void **device_ptr, size_t *pitch_bytes);
KC_API int kc_plane_upload_f32(kc_plane *p, const float *host, size_t host_pitch_bytes);
KC_API int kc_plane_download_f32(kc_plane *p, float *host, size_t host_pitch_bytes);

/* ========================================================================================== *
 * Images -- SlotImage, src/slot_image.rs:15-264
 * ========================================================================================== */
KC_API int kc_image_gray(kc_plane *p, kc_image **out);                 /* SlotImage::Gray */
KC_API int kc_image_rgba(kc_plane *const planes[4], kc_image **out);   /* SlotImage::Rgba */
KC_API int kc_image_retain(kc_image *img);
KC_API int kc_image_release(kc_image *img);
KC_API int kc_image_is_rgba(const kc_image *img, int *is_rgba);        /* slot_image.rs:123-139 */
KC_API int kc_image_size(const kc_image *img, kc_size *size);          /* slot_image.rs:116-121 */
KC_API int kc_image_plane(const kc_image *img, int channel, kc_plane **out); /* +1 reference */
KC_API int kc_image_from_value(kc_size size, float value, int rgba, kc_image **out); /* :28-64 */
KC_API int kc_image_as_type(const kc_image *img, int rgba, kc_image **out);          /* :212-256 */
KC_API int kc_image_materialize(kc_image *img);
/* deconstruct_image + read_slot_image, src/shared.rs:16-56,218-261: interleaved u8 with 1..4
 * channels -> RGBA planes (/255., missing R,G,B = 0, A = 1). */
KC_API int kc_image_from_u8(const uint8_t *host, uint32_t width, uint32_t height, int channels, kc_image **out);
/* to_u8 / to_u8_srgb, src/slot_image.rs:141-207: -> interleaved RGBA8 (width*height*4 bytes). */
KC_API int kc_image_to_u8(kc_image *img, int srgb, uint8_t *host_rgba8);
/* The same two as a PIPELINE for jobs that run one graph over many images: `depth` slots, each with a pinned host buffer for an
 * input image (width * height * channels bytes), one for an RGBA8 output image, and device staging for both; the copies run on
 * two streams of the pipe's own, ordered with the compute stream by events, so the upload of image k + 1 and the download of
 * image k - 1 overlap the evaluation of image k and no call waits on the host except kc_u8_pipe_wait_download.
 *   kc_u8_pipe_buffers        the slot's buffers: fill `*host_in` before kc_u8_pipe_upload, read `*host_out` after
 *                             kc_u8_pipe_wait_download (and before the slot's next kc_u8_pipe_download);
 *   kc_u8_pipe_upload         deconstruct_image of the slot's input buffer: `*out` (+1 ref) is usable at once (its planes are
 *                             ready in stream order).  Call it for the NEXT image after the current one's evaluation has been
 *                             enqueued -- the copy then runs during that evaluation;
 *   kc_u8_pipe_download       to_u8 (srgb = 0) / to_u8_srgb (1) of `img` into the slot's output buffer, asynchronously;
 *   kc_u8_pipe_wait_download  blocks until that buffer holds the image. */
KC_API int kc_u8_pipe_create(uint32_t width, uint32_t height, int channels, int depth, kc_u8_pipe **out);
KC_API int kc_u8_pipe_free(kc_u8_pipe *pipe);
KC_API int kc_u8_pipe_buffers(kc_u8_pipe *pipe, int slot, uint8_t **host_in, const uint8_t **host_out);
KC_API int kc_u8_pipe_upload(kc_u8_pipe *pipe, int slot, kc_image **out);
KC_API int kc_u8_pipe_download(kc_u8_pipe *pipe, int slot, kc_image *img, int srgb);
KC_API int kc_u8_pipe_wait_download(kc_u8_pipe *pipe, int slot);
KC_API int kc_image_from_f32(const float *const host_planes[], int n_planes, uint32_t width, uint32_t height, kc_image **out);
KC_API int kc_image_to_f32(kc_image *img, float *const host_planes[], int n_planes);
/* read_slot_image, src/shared.rs:218-261 (PNG only; decode on host, planes built on device). */
KC_API int kc_image_read_png(const char *path, kc_image **out);
KC_API int kc_image_write_png(kc_image *img, const char *path);        /* src/node/write.rs:5-21 */

/* ========================================================================================== *
 * Per-node operators -- the functions behind process_node_internal, src/node/node_type.rs:98-138.
 * Inputs arrive already resized and keyed by input slot (NULL = slot not connected), as after
 * resize_buffers + assign_slot_ids (node_type.rs:229-237,250-267).
 * ========================================================================================== */
/* calculate_size, src/shared.rs:61-139.  sizes[] in edge insertion order; slot_index = index of
 * the input SpecificSlot resolves to, or -1. */
KC_API int kc_calculate_size(int policy, const kc_size *sizes, int n, int slot_index, kc_size specific, kc_size *out);
/* image::imageops::resize per plane, call sites src/shared.rs:159-199. */
KC_API int kc_resize_image(kc_image *src, kc_size size, int filter, kc_image **out);
/* resize_buffers, src/shared.rs:141-216: images[] in edge insertion order, edges[] sorted by
 * input_slot (node_type.rs:230-231); keys[i] = (output node id, output slot id) of images[i]. */
KC_API int kc_resize_buffers(kc_image *const images[], const kc_edge keys[], int n, const kc_edge *edges_sorted,
                             int n_edges, int policy, uint32_t policy_slot, kc_size policy_size, int filter,
                             kc_image *out[]);
/* mix::process, src/node/mix.rs:51-134.  *out = NULL with KC_OK when the reference returns an
 * empty Vec (mixed Gray/Rgba after type matching, :126). */
KC_API int kc_mix_process(kc_image *left, kc_image *right, int mix_type, kc_image **out);
/* separate_rgba::process, src/node/separate_rgba.rs:38-69 */
KC_API int kc_separate_rgba_process(kc_image *input, kc_image *out[4]);
/* combine_rgba::process, src/node/combine_rgba.rs:14-97; first = slot_datas.get(0) (size source). */
KC_API int kc_combine_rgba_process(kc_image *const inputs[4], kc_image **out);
/* value::process, src/node/value.rs:14-26 */
KC_API int kc_value_process(float value, kc_image **out);
/* height_to_normal::process, src/node/height_to_normal.rs:16-77; *out = NULL when the reference
 * returns an empty Vec (no input / RGBA input). */
KC_API int kc_height_to_normal_process(kc_image *input, kc_image **out);

/* ========================================================================================== *
 * NodeGraph -- src/node_graph.rs:16-590
 * ========================================================================================== */
KC_API int kc_node_graph_new(kc_node_graph **out);                                  /* :25-31 */
KC_API int kc_node_graph_clone(const kc_node_graph *g, kc_node_graph **out);
KC_API int kc_node_graph_free(kc_node_graph *g);
KC_API int kc_node_graph_from_path(const char *path, kc_node_graph **out);          /* :33-46 */
KC_API int kc_node_graph_from_json(const char *json, kc_node_graph **out);          /* :104-107 */
KC_API int kc_node_graph_export_json(const kc_node_graph *g, const char *path);     /* :98-102 */
/* Serialises into buf (NUL-terminated); *needed = bytes required including the NUL. */
KC_API int kc_node_graph_to_json(const kc_node_graph *g, char *buf, size_t cap, size_t *needed);
KC_API int kc_node_graph_add_node(kc_node_graph *g, const kc_node_desc *node, uint32_t *node_id);       /* :315-320 */
KC_API int kc_node_graph_add_node_with_id(kc_node_graph *g, const kc_node_desc *node);                  /* :322-331 */
KC_API int kc_node_graph_connect(kc_node_graph *g, uint32_t output_node, uint32_t input_node, uint32_t output_slot, uint32_t input_slot); /* :416-446 */
KC_API int kc_node_graph_try_connect(kc_node_graph *g, uint32_t output_node, uint32_t input_node, uint32_t output_slot, uint32_t input_slot); /* :396-413 */
KC_API int kc_node_graph_remove_node(kc_node_graph *g, uint32_t node_id);                              /* :473-481 */
KC_API int kc_node_graph_remove_edge(kc_node_graph *g, kc_edge edge);                                  /* :462-471 */
KC_API int kc_node_graph_disconnect_slot(kc_node_graph *g, uint32_t node_id, int side, uint32_t slot_id); /* :496-515 */
KC_API int kc_node_graph_node_count(const kc_node_graph *g, uint32_t *count);
KC_API int kc_node_graph_node_ids(const kc_node_graph *g, uint32_t *ids, uint32_t cap, uint32_t *count); /* :125-127 */
KC_API int kc_node_graph_edges(const kc_node_graph *g, kc_edge *edges, uint32_t cap, uint32_t *count);
KC_API int kc_node_graph_input_slot_id_with_name(const kc_node_graph *g, const char *name, uint32_t *slot_id);  /* :271-276 */
KC_API int kc_node_graph_output_slot_id_with_name(const kc_node_graph *g, const char *name, uint32_t *slot_id); /* :278-283 */
KC_API int kc_node_graph_set_mix_type(kc_node_graph *g, uint32_t node_id, int mix_type);               /* :48-63 */
KC_API int kc_node_graph_set_image_node_path(kc_node_graph *g, uint32_t node_id, const char *path);    /* :65-83 */
/* Gives an Output node a new (de-collided) name; old_name (may be NULL) receives the previous one. */
KC_API int kc_node_graph_rename_output_node(kc_node_graph *g, uint32_t node_id, const char *new_name, char *old_name, size_t cap); /* :232-269 */

/* ========================================================================================== *
 * TextureProcessor / LiveGraph -- src/texture_processor.rs:18-115, src/live_graph.rs:63-645.
 * Evaluation is synchronous and stream-ordered: await_clean runs every dirty ancestor in
 * topological order on the calling thread (replaces engine::process_loop, src/engine.rs:25-312).
 * ========================================================================================== */
KC_API int kc_tex_pro_new(uint64_t memory_threshold, kc_tex_pro **out);              /* texture_processor.rs:34-56 */
KC_API int kc_tex_pro_free(kc_tex_pro *tp);
KC_API int kc_tex_pro_new_live_graph(kc_tex_pro *tp, kc_live_graph **out);           /* :58-64 */
KC_API int kc_live_graph_free(kc_live_graph *lg);
KC_API int kc_live_graph_set_flags(kc_live_graph *lg, int auto_update, int use_cache); /* live_graph.rs:71-72 */
KC_API int kc_live_graph_get_flags(const kc_live_graph *lg, int *auto_update, int *use_cache);
KC_API int kc_live_graph_set_node_graph(kc_live_graph *lg, const kc_node_graph *g);  /* :605-609 */
KC_API int kc_live_graph_node_graph(const kc_live_graph *lg, kc_node_graph **out_clone);
KC_API int kc_live_graph_add_node(kc_live_graph *lg, const kc_node_desc *node, uint32_t *node_id);      /* :426-433 */
KC_API int kc_live_graph_add_node_with_id(kc_live_graph *lg, const kc_node_desc *node);                 /* :435-444 */
KC_API int kc_live_graph_remove_node(kc_live_graph *lg, uint32_t node_id);                              /* :452-475 */
KC_API int kc_live_graph_connect(kc_live_graph *lg, uint32_t output_node, uint32_t input_node, uint32_t output_slot, uint32_t input_slot); /* :488-511 */
KC_API int kc_live_graph_remove_edge(kc_live_graph *lg, kc_edge edge);                                  /* :551-566 */
KC_API int kc_live_graph_disconnect_slot(kc_live_graph *lg, uint32_t node_id, int side, uint32_t slot_id); /* :568-594 */
KC_API int kc_live_graph_set_mix_type(kc_live_graph *lg, uint32_t node_id, int mix_type);               /* node_mut, :369-374 */
KC_API int kc_live_graph_rename_output_node(kc_live_graph *lg, uint32_t node_id, const char *new_name, char *old_name, size_t cap); /* :625-627 */
KC_API int kc_live_graph_set_resize(kc_live_graph *lg, uint32_t node_id, int policy, uint32_t policy_slot, kc_size policy_size, int filter);
KC_API int kc_live_graph_node_state(const kc_live_graph *lg, uint32_t node_id, int *state);             /* :244-250 */
KC_API int kc_live_graph_request(kc_live_graph *lg, uint32_t node_id);                                  /* :219-227 */
KC_API int kc_live_graph_prioritise(kc_live_graph *lg, uint32_t node_id);                               /* :229-237 */
/* await_clean_read / await_clean_write, :164-195: returns once node_id is Clean. */
KC_API int kc_live_graph_await_clean(kc_live_graph *lg, uint32_t node_id);
/* One scheduler pass: processes every Requested / Prioritised node (all non-clean nodes when
 * auto_update), src/engine.rs:128-183. */
KC_API int kc_live_graph_update(kc_live_graph *lg);
KC_API int kc_live_graph_slot_data(kc_live_graph *lg, uint32_t node_id, uint32_t slot_id, kc_image **out);   /* :415-420, +1 ref */
KC_API int kc_live_graph_slot_data_size(kc_live_graph *lg, uint32_t node_id, uint32_t slot_id, kc_size *size); /* :406-408 */
KC_API int kc_live_graph_slot_in_memory(kc_live_graph *lg, uint32_t node_id, uint32_t slot_id, int *in_memory); /* :410-412 */
KC_API int kc_live_graph_node_slot_ids(kc_live_graph *lg, uint32_t node_id, uint32_t *slot_ids, uint32_t cap, uint32_t *count); /* node_slot_datas, :389-404 */
KC_API int kc_live_graph_buffer_rgba(kc_live_graph *lg, uint32_t node_id, uint32_t slot_id, int srgb, uint8_t *host_rgba8); /* :93-95 */
KC_API int kc_live_graph_embed_slot_data_with_id(kc_live_graph *lg, kc_image *image, uint32_t slot_id, uint32_t embed_id); /* :324-341 */
KC_API int kc_live_graph_add_input_slot_data(kc_live_graph *lg, uint32_t node_id, uint32_t slot_id, kc_image *image);     /* :347-350 */
KC_API int kc_live_graph_changed_consume(kc_live_graph *lg, uint32_t *ids, uint32_t cap, uint32_t *count);               /* :156-160 */
KC_API int kc_live_graph_output_ids(const kc_live_graph *lg, uint32_t *ids, uint32_t cap, uint32_t *count);              /* :621-623 */
KC_API int kc_live_graph_node_ids(const kc_live_graph *lg, uint32_t *ids, uint32_t cap, uint32_t *count);                /* :629-631 */
KC_API int kc_live_graph_edges(const kc_live_graph *lg, kc_edge *edges, uint32_t cap, uint32_t *count);                  /* :633-635 */
/* Base directory that relative Image / Write paths resolve against (the reference resolves them
 * against the process's working directory). */
KC_API int kc_live_graph_set_base_dir(kc_live_graph *lg, const char *dir);

/* ========================================================================================== *
 * Multi-GPU: one process per GPU, every process holds the same graph.  The reference runs every ready
 * node on its own thread and needs nothing but the parents' slot data to do so (src/engine.rs:213-275,
 * :288); here the same rule lets independent branches run on different GPUs.  kc_live_graph_partition
 * derives, from the graph alone (so every rank computes the same answer without communicating), which rank
 * evaluates which ancestor of `root` and which slots cross a rank boundary -- or that every rank takes a band of rows.  The
 * library moves the slots itself (kc_comm_*, kc_live_graph_evaluate_partitioned below); a host with a transport of its own
 * can do it through kc_plane_device_ptr / kc_plane_alloc + kc_image_gray / kc_image_rgba and kc_live_graph_import_slot_data.
 * INTEGRATION.md shows both.
 * ========================================================================================== */
typedef enum kc_partition_policy {
    KC_PARTITION_AUTO = 0,   /* the cheapest of {one GPU, branches with transfers charged, row bands + gather of the result},
                              * priced in one unit (one RGBA 4096^2 slot over xGMI ~ 12 fused Mix chains): small graphs stay
                              * on one GPU, wide pointwise graphs (BASELINE config #4) go by rows */
    KC_PARTITION_SPREAD = 1, /* branches, transfers not charged: independent branches fill all ranks */
    KC_PARTITION_BANDS = 2   /* row bands (KC_ERR_UNSUPPORTED when the band walk does not take the graph or the sources'
                              * sizes are not known yet) */
} kc_partition_policy;
typedef enum kc_plan_kind {
    KC_PLAN_SINGLE = 0,   /* everything on the home rank, nothing moves */
    KC_PLAN_BRANCHES = 1, /* nodes placed per rank, `transfers` cross rank boundaries */
    KC_PLAN_BANDS = 2     /* every rank evaluates rows [y0, y1) of the requested node (kc_partition_bands); the pointwise
                           * nodes (src/node/mix.rs:136-192) need no exchange at all, the finished bands are gathered on the
                           * home rank unless kc_partition_set_gather(plan, 0) */
} kc_plan_kind;
typedef struct kc_band_range { int32_t y0, y1; } kc_band_range;
typedef enum kc_node_kind {
    KC_KIND_SOURCE = 0,     /* Embed / Image / Input*: data somebody put there; lives on `rank` */
    KC_KIND_REPLICATED = 1, /* no source among its ancestors (Value nodes and constants built from them):
                             * evaluated on every rank that needs it, never sent; rank = -1 */
    KC_KIND_COMPUTE = 2     /* evaluated on `rank` only */
} kc_node_kind;
typedef struct kc_placement { uint32_t node_id; int32_t rank; int32_t component; int32_t kind; } kc_placement;
/* One slot moving from the rank that produced it to ONE consumer rank; a slot consumed on several ranks (a
 * broadcast) appears once per destination, consecutively.  Transfers are listed in the order every rank must
 * work through them; `level` = how many rank boundaries the data has crossed before (0 for branch results). */
typedef struct kc_transfer { uint32_t node_id, slot_id; int32_t src_rank, dst_rank; int32_t level; } kc_transfer;
KC_API int kc_live_graph_partition(kc_live_graph *lg, uint32_t root_node_id, int world_size, int policy, kc_partition **out);
KC_API int kc_partition_free(kc_partition *p);
KC_API int kc_partition_info(const kc_partition *p, int *world_size, int *home_rank, int *levels);
/* Ancestors of the root (root included) in topological order, with their placement. */
KC_API int kc_partition_nodes(const kc_partition *p, kc_placement *out, uint32_t cap, uint32_t *count);
KC_API int kc_partition_transfers(const kc_partition *p, kc_transfer *out, uint32_t cap, uint32_t *count);
/* Which of the three a plan is, and what KC_PARTITION_AUTO compared (estimated times in units of one fused RGBA Mix chain over
 * the image; `bands` < 0: no band plan exists).  Any of the out pointers may be NULL. */
KC_API int kc_partition_kind(const kc_partition *p, int *kind, double *est_single, double *est_branches, double *est_bands);
/* KC_PLAN_BANDS: the rows of the requested node per rank (`count` = world size) and the node's full size.  A rank holds the rows
 * kc_live_graph_band_source_rows names for its band of every source (kc_live_graph_embed_slot_data_band), or whole sources. */
KC_API int kc_partition_bands(const kc_partition *p, kc_band_range *out, uint32_t cap, uint32_t *count, uint32_t *full_width, uint32_t *full_height);
/* KC_PLAN_BANDS: gather = 0 leaves every rank's band where it is (kc_live_graph_evaluate_partitioned then returns the band on
 * every rank); the default, 1, assembles the image on the home rank. */
KC_API int kc_partition_set_gather(kc_partition *p, int gather);
/* Stores `image` (+1 ref) as slot `slot_id` of `node_id` and marks the node Clean, exactly as the engine does with
 * the result of a finished node (src/engine.rs:34-57): the receiving side of a transfer. */
KC_API int kc_live_graph_import_slot_data(kc_live_graph *lg, uint32_t node_id, uint32_t slot_id, kc_image *image);

/* The exchange itself, inside the library (csrc/comm.cpp): the slots a branch plan cuts, and the finished bands of a band plan,
 * move between the processes of one node.  Descriptions of slots (size; constant planes travel as scalars, aliased planes once)
 * go through a shared-memory mailbox, host to host; the planes go over one of two wires, chosen by rank 0 when it makes the id
 * (environment KC_COMM_TRANSPORT):
 *   "ipc" (default)  the consumer maps the producer's planes (hipIpcOpenMemHandle), one stream per peer waits ON THE DEVICE for a
 *                    counter the producer's stream writes behind its kernels, copies (hipMemcpyAsync: the DMA engines over xGMI)
 *                    and acknowledges the same way; works between processes sharing one GPU too;
 *   "rccl"           ncclSend / ncclRecv, all transfers of a level in one group (librccl bound at first use; a process that never
 *                    asks for it needs no RCCL).
 * No host thread waits for plane data with either.  One communicator per process:
 *   kc_comm_unique_id   rank 0 fills `id` (KC_COMM_ID_BYTES); the host passes it to every rank by whatever channel it has
 *                       (a file, a pipe, MPI, torch.distributed's store ...);
 *   kc_comm_init        collective over all ranks (at most 16), after kc_init; a failure is a status (nothing is retried);
 *   kc_live_graph_exchange  works through `transfers` level by level -- every rank passes the same list, e.g. what
 *                       kc_partition_transfers returns; transfers of one level must not depend on each other.  Per level the
 *                       producer's rank evaluates the node (kernels enqueued, nobody waits) and posts the description; a
 *                       consumer's rank allocates planes, receives into them and hands the slot over exactly as
 *                       kc_live_graph_import_slot_data does, its compute stream waiting for the transfer on the device.
 *                       Consecutive entries of one slot are one multi-destination send.  An entry from a rank to itself is legal
 *                       (the slot is replaced by the copy that came back).
 *   kc_comm_gather_bands  every rank passes its band (rows y0 .. of an image `full_height` rows high, e.g. what
 *                       kc_live_graph_evaluate_band returned); on `home_rank`, `*out` (+1 ref) is the assembled image (the
 *                       bands must tile it), NULL elsewhere.  Each band travels over its own link, straight to its row offset.
 *   kc_live_graph_evaluate_partitioned  the whole evaluation of a plan: KC_PLAN_SINGLE / KC_PLAN_BRANCHES -- what this rank can
 *                       compute before anything arrives is enqueued first (it overlaps the transfers), then the exchange, then
 *                       `root` on the plan's home rank: `*out` (+1 ref) is the root's result there and NULL on the other ranks;
 *                       KC_PLAN_BANDS -- this rank's rows of `root`, then the gather (or, after kc_partition_set_gather(plan, 0),
 *                       `*out` = the band on every rank).
 * Any failure on any rank makes every host-side wait of every rank fail (they also time out: KC_COMM_TIMEOUT_S seconds, default
 * 120); the communicator is unusable afterwards.  The readiness rule that makes all this correct is the reference's: a node needs
 * nothing but its parents' slot data (src/engine.rs:213-275). */
#define KC_COMM_ID_BYTES 256
KC_API int kc_comm_unique_id(void *id);
KC_API int kc_comm_init(int rank, int world_size, const void *id);
KC_API int kc_comm_destroy(void);
KC_API int kc_comm_info(int *rank, int *world_size);  /* 0, 0 without a communicator */
KC_API int kc_comm_transport(char *buf, size_t cap);  /* "ipc", "rccl" or "" */
KC_API int kc_comm_stats(uint64_t *planes_sent, uint64_t *planes_received, uint64_t *bytes_sent);
KC_API int kc_live_graph_exchange(kc_live_graph *lg, const kc_transfer *transfers, uint32_t count);
KC_API int kc_comm_gather_bands(kc_image *band, int32_t y0, uint32_t full_height, int home_rank, kc_image **out);
KC_API int kc_live_graph_evaluate_partitioned(kc_live_graph *lg, const kc_partition *plan, uint32_t root_node_id, kc_image **out);

/* Row bands: rows [y0, y1) of a node's result without computing the rest -- the data-level way to put several GPUs on one
 * graph.  Every pixel goes through the same operations as in the whole-image evaluation (process_node, src/node/
 * node_type.rs:213-248), so the bands of all ranks, stacked, equal it bit for bit.  Pointwise nodes need the same rows
 * of their inputs; HeightToNormal one more row on top (toroidal: the band that starts at row 0 needs the LAST row,
 * src/node/process_shared.rs:31-65); an implicit resize the rows its vertical taps read (src/shared.rs:159-199).  The
 * evaluation widens every intermediate band by those halo rows and computes them redundantly: no exchange between ranks.
 * Graph nodes (src/node/graph.rs:14-51) are expanded into their graphs before the walk (the resize the Graph node applies to
 * its inputs becomes a SpecificSize pass-through in front of every inner Input node), nested ones too; Write nodes are
 * not supported here.
 *   kc_live_graph_evaluate_band: `*out` (+1 ref) is an image of (y1 - y0) rows, resident on return.
 *   kc_live_graph_band_source_rows: which rows of every SOURCE (Embed / Image / Input*) that evaluation reads, so that
 *     sources which are themselves sharded by rows can be loaded with exactly their halo.  y0 may be negative: row -1 is
 *     the image's last row (the wrap).  Host only, no device needed.
 *   kc_live_graph_embed_slot_data_band: like kc_live_graph_embed_slot_data_with_id, but `image` holds only logical rows
 *     band_y0 .. band_y0 + rows - 1 of an image `full_height` rows high (band_y0 < 0: the wrapped rows come first). */
typedef struct kc_band_rows { uint32_t node_id; int32_t y0, y1; uint32_t width, height; } kc_band_rows;
KC_API int kc_live_graph_evaluate_band(kc_live_graph *lg, uint32_t node_id, uint32_t slot_id, int32_t y0, int32_t y1, kc_image **out);
KC_API int kc_live_graph_band_source_rows(kc_live_graph *lg, uint32_t node_id, int32_t y0, int32_t y1, kc_band_rows *rows, uint32_t cap, uint32_t *count);
KC_API int kc_live_graph_embed_slot_data_band(kc_live_graph *lg, kc_image *image, uint32_t slot_id, uint32_t embed_id, int32_t band_y0, uint32_t full_height);

#ifdef __cplusplus
}
#endif
#endif /* KANTER_CORE_AMD_H */
