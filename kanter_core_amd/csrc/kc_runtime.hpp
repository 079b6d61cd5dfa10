// Host runtime declarations: context, HBM plane pool, planes / images, node operators, graphs.
#pragma once

#include <cstring>
#include <deque>
#include <set>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <unordered_set>

#include "kc_internal.hpp"

// ------------------------------------------------------------------------------------------
// Opaque C-ABI types
// ------------------------------------------------------------------------------------------
namespace kc {
struct Chain;
struct ChainLink;
}

// Fixed-size objects the evaluator creates and drops by the hundred per evaluation (a plane, a chain link and an
// image per node and channel) come from free lists: released together when a chain has run, they overflow
// malloc's small per-thread cache and took its locked fast-bin path -- a third of the host time of a 256 x 256
// evaluation (profiles/r02_host_samples.txt).  The lists are guarded by Context::mu, which every creator and
// releaser of these objects holds (see kc_plane::refs); a list keeps at most kMaxFree blocks.  Sanitizer builds
// use plain new / delete so that they still see every use after free.
namespace kc {
#if defined(__SANITIZE_ADDRESS__) || defined(__SANITIZE_THREAD__)
#define KC_NO_OBJECT_POOL 1
#elif defined(__has_feature)
#if __has_feature(address_sanitizer) || __has_feature(thread_sanitizer)
#define KC_NO_OBJECT_POOL 1
#endif
#endif
template <class T> struct Pooled {
#ifndef KC_NO_OBJECT_POOL
    struct FreeList {
        struct Block { Block *next; };
        static constexpr size_t kMaxFree = 1u << 14;
        Block *head = nullptr;
        size_t n = 0;
        bool closed = false;  // the process is exiting: objects released by later exit handlers go straight to the heap
        ~FreeList()
        {
            closed = true;
            while (head) {
                Block *b = head;
                head = b->next;
                ::operator delete(b);
            }
            n = 0;
        }
    };
    static FreeList &list()
    {
        static FreeList fl;
        return fl;
    }
    static void *operator new(size_t bytes)
    {
        FreeList &fl = list();
        if (bytes == sizeof(T) && fl.head) {
            typename FreeList::Block *b = fl.head;
            fl.head = b->next;
            --fl.n;
            return b;
        }
        return ::operator new(bytes < sizeof(typename FreeList::Block) ? sizeof(typename FreeList::Block) : bytes);
    }
    static void operator delete(void *q, size_t bytes)
    {
        FreeList &fl = list();
        if (bytes == sizeof(T) && fl.n < FreeList::kMaxFree && !fl.closed) {
            typename FreeList::Block *b = static_cast<typename FreeList::Block *>(q);
            b->next = fl.head;
            fl.head = b;
            ++fl.n;
            return;
        }
        ::operator delete(q);
    }
#endif
};
}  // namespace kc

// One channel.  MEM: pitched f32 in HBM.  CONST: broadcast scalar (what the reference holds as
// vec![v; n]).  LAZY: a pointwise Mix chain that has not been run yet; forcing it launches the
// fused chain kernel and turns the plane into MEM.  RESIZE: `rz_src` resampled to w x h with
// `rz_filter`, not run yet: a Mix chain that consumes it resamples inside its own kernel
// (resize_chain_kernel), anything else forces it through the plain resize kernel.
struct kc_plane : kc::Pooled<kc_plane> {
    enum Kind { MEM = 0, CONST = 1, LAZY = 2, RESIZE = 3 };
    // Reference counts are plain integers: every C-ABI entry that creates, retains or releases a plane or an
    // image holds Context::mu (c_api.cpp), as the evaluator always did -- an evaluation touches them some
    // thirty times per node, and the locked read-modify-writes were a quarter of its host time.
    int refs = 1;
    uint32_t w = 0, h = 0;
    Kind kind = MEM;
    float *dptr = nullptr;
    size_t pitch = 0;  // bytes
    size_t bytes = 0;  // pool block size (owned planes)
    bool owned = false;
    float cval = 0.0f;
    kc::ChainLink *link = nullptr;  // LAZY: the last step and what it continues (see ChainLink)
    kc::Chain *chain = nullptr;     // LAZY: the flattened program, built when the plane is forced
    kc_plane *view_of = nullptr;  // MEM, not owned: rows of this (retained) plane, see bands.cpp
    kc_plane *rz_src = nullptr;  // retained; MEM
    int rz_filter = 0;
};

// SlotImage, src/slot_image.rs:15-19
struct kc_image : kc::Pooled<kc_image> {
    int refs = 1;  // under Context::mu, like kc_plane::refs
    int n = 0;  // 1 = Gray, 4 = Rgba
    kc_plane *planes[4] = { nullptr, nullptr, nullptr, nullptr };
    bool is_rgba() const { return n == 4; }
    uint32_t w() const { return planes[0]->w; }
    uint32_t h() const { return planes[0]->h; }
};

namespace kc {

struct ChainStep {
    uint8_t code;       // ChainCode
    kc_plane *operand;  // retained; MEM or CONST -- or, in a link, LAZY: a second chain joined in (plane_mix); in a flat
                        // Chain that one is expanded and the combining step's operand is saved_value_marker()
    uint8_t level = 0;  // flat Chain only: which saved value a CH_SAVE_LOAD writes / a step on saved_value_marker() reads
};
kc_plane *saved_value_marker();

struct Chain : Pooled<Chain> {
    kc_plane *start = nullptr;  // retained; MEM, CONST or RESIZE
    std::vector<ChainStep> steps;
    std::vector<kc_plane *> joined;  // retained; the LAZY planes whose chains were expanded into `steps` (CH_SAVE_LOAD)
    ~Chain();
};

// A lazy plane is "the plane it continues, plus one step": extending a chain is O(1) and consumers of
// the same prefix share it (a 64-node linear graph used to copy 3 x 64 x 32 steps per evaluation).
// The flat Chain is built from the links when the plane is forced; a `prev` that has been forced in
// the meantime is resident and simply becomes the start.
struct ChainLink : Pooled<ChainLink> {
    kc_plane *prev = nullptr;   // retained; the LAZY plane continued, or nullptr
    kc_plane *start = nullptr;  // retained; the chain's first value when prev == nullptr
    ChainStep step{};           // operand retained
    uint32_t length = 1;        // RECORDS up to and including this step (an upper bound once a prev was forced): a step that
                                // chain_fill folds into its predecessor's record (CH_*_INV) does not count
    bool fused = false;         // this step is such a folded one
    uint8_t saved = 0;          // saved values the chain needs at once: 0 = no joined chain, 1 = joins of plain chains, ...
    int n_in = 0;               // distinct MEM / RESIZE planes the whole chain reads (same bound)
    // Their identities BY VALUE (device pointer + pitch of a resident plane, address of a deferred resize):
    // the planes themselves are kept alive by the links that use them, which may be gone once a prefix
    // has been forced, so these are never dereferenced.  A stale entry only makes the bound more cautious.
    struct InKey {
        const void *p = nullptr;
        size_t q = 0;
        bool operator==(const InKey &o) const { return p == o.p && q == o.q; }
    };
    InKey ins[KC_CHAIN_MAX_IN + 1];
    ~ChainLink();
};

struct TapsHost {
    std::vector<uint32_t> left, count;
    std::vector<float> w;
    uint32_t stride = 1;     // max taps of any output index
    uint32_t min_count = 1;  // min taps of any output index
    // Integer-ratio down-sampling: output indices [reg_a, reg_b) all have reg_ages * reg_ratio taps, windows reg_ratio
    // apart and bit-identical weights (resize_poly_kernel); reg_ratio == 0: no such range.
    uint32_t reg_a = 0, reg_b = 0, reg_ages = 0, reg_ratio = 0;
    // Integer-ratio up-sampling (upsample.h): the table has that structure (up_axis_build checks it bit for bit);
    // up_rows = the class rows, up.cls points at their copy in HBM once the table is uploaded.
    bool up_ok = false;
    UpAxis up{};
    std::vector<float> up_rows, up_qrows;  // class rows per output / per column quad (tap-major)
    // resize_down2_kernel (down2.hip; built by down2_build for down-sampling axes, d2_want):
    //   as the VERTICAL table   d2_vrec: d2_nc records of KC_DOWN2_REC dwords per group of four output rows (kc_internal.hpp);
    //                           d2_nc == 0: some group's windows span more than KC_DOWN2_MAX_CHUNKS x 16 source rows (wrapped
    //                           rows of a band table, very large ratios)
    //   as the HORIZONTAL table d2_hw: the weights with rows padded to d2_hstride (a multiple of 4, +0.0 beyond a column's own
    //                           taps); d2_tile_w: the strip width whose widest source window is 64 quads (0: none)
    std::vector<uint32_t> d2_vrec, d2_strips;  // d2_strips: (first source column & ~3, source quads) per strip of d2_tile_w columns
    std::vector<float> d2_hw;
    bool d2_want = false;  // a down-sampling axis with windows of at least 4 taps
    uint32_t d2_nc = 0, d2_hstride = 0, d2_tile_w = 0;
    uint32_t p2_tile_w = 0;  // as the HORIZONTAL table of resize_poly2_kernel: the strip width (<= 128) whose widest window is 64 quads
    const uint32_t *d2_vrec_dev = nullptr, *d2_strips_dev = nullptr;
    const float *d2_hw_dev = nullptr;
};

struct TapsEntry {
    uint64_t last_use = 0;  // band tables only: LRU clock (Context::band_taps_clock)
    TapsHost host;
    void *dev_block = nullptr;
    size_t dev_bytes = 0;
    TapsDev dev{};
};

// What chain_launch leaves behind while an evaluation is being recorded for replay (replay.cpp): one record per launch.
struct ReplayLaunch {
    ChainProgram prog;  // as launched: input and output pointers of the recorded run
    int batch = 0, mode = 0;
    uint32_t in_refs[KC_CHAIN_MAX_IN] = {};
    uint32_t w = 0, h = 0;
    kc_plane *planes[KC_CHAIN_MAX_BATCH] = { nullptr, nullptr, nullptr, nullptr };  // the lazy planes the launch made resident
    // filled when the recording is closed: input (b, k) is output channel in_ch of recorded launch in_from (-1: a plane that
    // existed before the evaluation and is held by the recording)
    int in_from[KC_CHAIN_MAX_BATCH][KC_CHAIN_MAX_IN];
    int in_ch[KC_CHAIN_MAX_BATCH][KC_CHAIN_MAX_IN];
};
struct ReplayCapture {
    bool ok = true;
    std::vector<ReplayLaunch> launches;
};

struct Context {
    std::recursive_mutex mu;
    ReplayCapture *capture = nullptr;
    uint64_t band_taps_clock = 0;
    bool inited = false;
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool fusion = true;
    bool replay = true;  // an evaluation that repeats the recorded one is replayed without the walk (kc_set_option("replay", 0); env KC_REPLAY)
    int down2 = 1;       // resize_down2_kernel: 0 off, 1 except where resize_poly_kernel runs at ratio 4 or 8, 2 there too (kc_set_option("down2"); env KC_DOWN2)
    int down2_by_rows = -1;   // resize_down2_kernel's job order: four strips of one row group per workgroup, XCDs in eighths row by row;
                              // -1: where the row groups' windows span several chunks (kc_set_option("down2_by_rows"); env KC_DOWN2_BY_ROWS)
    int poly2 = 1;            // integer-ratio down-sampling through resize_poly2_kernel (kc_set_option("poly2"); env KC_POLY2); 0: resize_poly_kernel / down2 as before
    int poly2_min_ratio = 8;  // ... from this vertical ratio on (kc_set_option("poly2_min_ratio"); env KC_POLY2_MIN_RATIO): where it measures faster
    bool plain_chains = false;  // set during a graph's first evaluation: chains as the interpreter runs them (4 planes, no joins)
    bool wide = true;    // chains of up to KC_CHAIN_MAX_IN input planes (compiled kernels only); 0: KC_CHAIN_INTERP_IN as before (kc_set_option("wide"); env KC_WIDE)
    bool join = true;    // a Mix of two unevaluated chains keeps both in one program (kc_set_option("join", 0); env KC_JOIN)
    bool chain1 = true;  // one-step programs run the ahead-of-time kernels of chain1.hip (kc_set_option("chain1", 0): interpreter / specialiser, A/B and tests)
    int link_gbps = 153;   // one xGMI link, what the planner charges a transfer with (kc_set_option("link_gbps"))
    int hbm_gbps = 6100;   // what a streaming kernel gets from HBM with nothing in the Infinity Cache (kc_set_option("hbm_gbps"))
    int cache_budget_mb = 208;  // what a launch may leave cacheable: 13/16 of the 256 MB Infinity Cache of an MI355X, the share that measured best (profiles/r03_tilecopy4.txt); kc_set_option("cache_budget_mb") / KC_CACHE_BUDGET_MB for another part
    int cache_policy = 1;  // 1: launches whose streams exceed the Infinity Cache mark them nontemporal (cache_policy_mask); 0: plain loads / stores (KC_CACHE_POLICY, kc_set_cache_policy)
    int max_blocks = 4096;
    int chain_unroll = 0;  // float4 per thread per decode in the chain kernel; 0 = heuristic (KC_CHAIN_UNROLL)
    int resize_mode = 0;  // 0 auto (tiled single pass when a tile fits in LDS), 1 no resize_poly_kernel, 2 no resize_down_kernel either (A/B), 3 two passes through HBM only, 4 auto without the integer-ratio up-sampling kernels (KC_RESIZE_MODE, kc_set_resize_mode)
    int resize_tile_w = 0;  // > 0: force this tile width (KC_RESIZE_TILE_W, tuning only)
    int resize_tile_h = 0;  // > 0: force this tile height for 256-wide tiles (KC_RESIZE_TILE_H, tuning only)
    std::multimap<size_t, void *> free_blocks;
    uint64_t bytes_in_use = 0, bytes_cached = 0, launches = 0;
    std::map<std::string, uint64_t> counters;  // named event counts (kc_stats_counter): which kernel family ran, transfers ...
    uint64_t alg_bytes = 0;  // algorithmic HBM bytes of every kernel launched so far (DESIGN.md section 3's per-kernel figures)
    std::map<std::tuple<uint32_t, uint32_t, int>, TapsEntry> taps;
    std::map<std::tuple<uint32_t, uint32_t, int, int32_t, int32_t, int32_t>, TapsEntry> band_taps;  // row-band vertical tables
    // Within one graph evaluation the same plane resized to the same size with the same filter is
    // computed once (the reference resamples it per consuming node, src/shared.rs:152-207; planes
    // are immutable, so the result is identical).  Both planes are retained while memoised.
    int memo_depth = 0;
    std::map<std::tuple<kc_plane *, uint32_t, uint32_t, int>, kc_plane *> resize_memo;
    // image_from_u8's pinned staging ring: the caller's pixels are copied into a slot, the slot is DMA'd and the
    // call returns without waiting for the stream; a slot is reused once the event recorded behind its copy has fired.
    struct UploadSlot {
        void *host = nullptr;
        size_t bytes = 0;
        hipEvent_t copied = nullptr;
    };
    static constexpr int kUploadSlots = 4;
    static constexpr size_t kUploadSlotMax = (size_t)4 << 20;  // larger images keep the synchronous path
    UploadSlot upload_ring[kUploadSlots];
    int upload_next = 0;
};

// RAII scope for the resize memo (nested Graph nodes share the outermost scope).
struct ResizeMemoScope {
    ResizeMemoScope();
    ~ResizeMemoScope();
};

Context &ctx();
void set_error(const std::string &msg);
const std::string &last_error();
int hip_fail(hipError_t e, const char *what);  // records the message, returns KC_ERR_HIP
int need_init();                               // KC_OK or KC_ERR_NO_DEVICE

// Host-side phase timers for tuning (build with KC_HOST_PROFILE=1 python -m kanter_core_amd.build --force): each
// KC_PROF("name") scope adds its TSC cycles to a named counter, printed at kc_shutdown.  Compiled out otherwise.
#ifdef KC_HOST_PROFILE
struct ProfScope {
    const char *name;
    unsigned long long t0;
    explicit ProfScope(const char *n) : name(n), t0(__builtin_ia32_rdtsc()) {}
    ~ProfScope();
};
void prof_report();
#define KC_PROF(name) kc::ProfScope _kc_prof_scope(name)
#else
#define KC_PROF(name) (void)0
#endif

#ifdef KC_HOST_SAMPLE
void sampler_start();   // SIGPROF sampling of the host side, see runtime.cpp
void sampler_report();
#endif

#define KC_HIP(call)                                           \
    do {                                                       \
        hipError_t _e = (call);                                \
        if (_e != hipSuccess) return kc::hip_fail(_e, #call);  \
    } while (0)
#define KC_TRY(expr)                   \
    do {                               \
        int _s = (expr);               \
        if (_s != KC_OK) return _s;    \
    } while (0)

// ---- pool / planes (runtime.cpp) ----
int pool_alloc(size_t bytes, void **out);
void pool_free(void *p, size_t bytes);
int pool_trim();

int plane_new_mem(uint32_t w, uint32_t h, kc_plane **out);
kc_plane *plane_new_const(uint32_t w, uint32_t h, float v);
inline void plane_retain(kc_plane *p)
{
    if (p) ++p->refs;
}
void plane_release(kc_plane *p);
int plane_force(kc_plane *p);                           // LAZY -> MEM (CONST stays CONST)
int planes_force(kc_plane *const *planes, int n);       // batches chains that share a program
int plane_materialize(kc_plane *p);                     // LAZY/CONST -> MEM
// l op r for one channel (src/node/mix.rs:136-192): lazy chain, constant fold or immediate kernel.
int plane_mix(int mix_type, kc_plane *l, kc_plane *r, kc_plane **out);
// Batches the forces plane_mix would do one plane at a time (the channels of one Mix node).
int planes_mix_prepare(kc_plane *const *ls, kc_plane *const *rs, int n);
Operand plane_operand(const kc_plane *p);               // MEM or CONST only

kc_image *image_new(int n, kc_plane *const *planes);    // retains the planes
inline void image_retain(kc_image *img)
{
    if (img) ++img->refs;
}
void image_release(kc_image *img);
int image_force(kc_image *img);

// ---- node operators (ops.cpp) ----
int image_from_value(kc_size size, float v, bool rgba, kc_image **out);
int image_as_type(kc_image *img, bool rgba, kc_image **out);
int image_from_u8(const uint8_t *host, uint32_t w, uint32_t h, int channels, kc_image **out);
int image_to_u8(kc_image *img, bool srgb, uint8_t *host);
int calculate_size(int policy, const kc_size *sizes, int n, int slot_index, kc_size specific, kc_size *out);
int mix_process(kc_image *left, kc_image *right, int mix_type, kc_image **out);
int separate_process(kc_image *in, kc_image *out[4]);
int combine_process(kc_image *const in[4], kc_image **out);
int value_process(float v, kc_image **out);
int height_to_normal_process(kc_image *in, kc_image **out);
int height_to_normal_band(kc_image *in_with_halo, uint32_t full_h, kc_image **out);  // rows + 1 input rows, halo first

// ---- resize (resize.cpp) ----
int build_taps_host(uint32_t in_n, uint32_t out_n, int filter, TapsHost &t);
int resize_image(kc_image *src, kc_size size, int filter, kc_image **out);
int resize_force(kc_plane *p);  // RESIZE -> MEM through the plain resize kernel
int resize_planes_band(kc_plane *const *srcs, int n, int32_t src_y0, uint32_t src_h_full, kc_size dst_full, int32_t a, int32_t b,
                       int filter, kc_plane **outs);
int resize_force_many(kc_plane *const *planes, int n);  // same; equal resamples share launches
// Runs a chain whose operands include ONE resampled plane per channel inside the resize kernel
// (phase 2 feeds the chain program).  *launched = false when the case is not eligible (taps not
// in registers, tile does not fit LDS, program too long): the caller forces the RESIZE planes.
int chain_resize_launch(const ChainProgram &P, int batch, int mode, kc_plane *const *sampled, bool *launched);

// ---- run-time specialisation of the chain kernel (specialize.cpp) ----
hipError_t launch_chain_specialized(const ChainProgram &P, int batch, hipStream_t s, bool *launched);
// Which streams of a launch are marked nontemporal: `in_bytes` / `out_bytes` = what the launch reads from its n_resident
// full-size input planes / writes, summed over its channels.  Returns the ChainProgram::nt_mask bits.
uint32_t cache_policy_mask(uint64_t in_bytes, uint64_t out_bytes, uint32_t n_resident);
uint32_t chain_cache_policy(const uint32_t *refs, uint32_t n, uint64_t stream_bytes, uint64_t out_bytes);
hipError_t chain_dispatch(ChainProgram &P, int batch, int mode, uint32_t w, uint32_t h, size_t out_pitch_bytes);
hipError_t launch_upsample_chain_specialized(const ChainProgram &P, int batch, const UpsampleArgs &U, hipStream_t s, bool *launched);
int specialize_set_mode(int mode, int after);  // 0 off, 1 background compile after `after` sightings, 2 compile at once
int specialize_get_mode();
void specialize_wait();
void specialize_stats(uint64_t *compiled, uint64_t *failed, uint64_t *launches, uint64_t *pending);
std::string specialize_last_log();
void specialize_shutdown();
std::string specialize_source(const ChainProgram &P, const UpsampleArgs *U = nullptr);
int specialize_compile_only(const ChainProgram &P, std::string *log, const UpsampleArgs *U = nullptr);
// code objects that outlive the process (specialize.cpp, "code objects that outlive the process")
int kernel_cache_set_dir(const char *dir);
void kernel_cache_stats(uint64_t *hits, uint64_t *rejected, uint64_t *written, uint64_t *kernels_loaded);
bool kernel_cache_populated();
int kernel_cache_precompile(const uint32_t *words, uint32_t n_ops, uint32_t n_in, int32_t start_src, bool flat, uint32_t nt_mask,
                            uint32_t up_taps, bool up_wide, const char *dir, std::string *log);

// ---- the u8 boundary as a pipeline (u8pipe.cpp) ----
int u8_pipe_create(uint32_t w, uint32_t h, int channels, int depth, kc_u8_pipe **out);
int u8_pipe_free(kc_u8_pipe *p);
int u8_pipe_buffers(kc_u8_pipe *p, int slot, uint8_t **host_in, const uint8_t **host_out);
int u8_pipe_upload(kc_u8_pipe *p, int slot, kc_image **out);
int u8_pipe_download(kc_u8_pipe *p, int slot, kc_image *img, bool srgb);
int u8_pipe_wait_download(kc_u8_pipe *p, int slot);

// ---- png / json (png.cpp, json.cpp) ----
int png_read(const std::string &path, std::vector<uint8_t> &px, uint32_t &w, uint32_t &h, int &channels);
int png_write_rgba8(const std::string &path, const uint8_t *px, uint32_t w, uint32_t h);

// ------------------------------------------------------------------------------------------
// Graph model: src/node_graph.rs, src/node/mod.rs:113-123, src/edge.rs
// ------------------------------------------------------------------------------------------
struct NodeGraph;

struct Node {
    uint32_t node_id = 0;
    int type = KC_NODE_VALUE;
    int mix_type = KC_MIX_ADD;
    float value = 0.0f;
    uint32_t embed_id = 0;
    std::string text;                  // Input/Output name, Image/Write path
    std::shared_ptr<NodeGraph> graph;  // Graph node
    int policy = KC_POLICY_MOST_PIXELS;
    uint32_t policy_slot = 0;
    kc_size policy_size{ 0, 0 };
    int filter = KC_FILTER_TRIANGLE;

    bool is_input() const { return type == KC_NODE_INPUT_GRAY || type == KC_NODE_INPUT_RGBA; }
    bool is_output() const { return type == KC_NODE_OUTPUT_GRAY || type == KC_NODE_OUTPUT_RGBA; }
};

enum SlotType { SLOT_GRAY = 0, SLOT_RGBA = 1, SLOT_GRAY_OR_RGBA = 2 };
struct Slot {
    std::string name;
    uint32_t slot_id;
    int slot_type;
};

// Look-up tables over NodeGraph::nodes / edges, rebuilt on first use after a change: the evaluator asks
// "which node is this id", "which edges enter / leave this node" several times per node, and linear
// scans made an evaluation quadratic in the node count.
struct GraphIndex {
    uint64_t version = ~0ull;
    size_t n_nodes = 0, n_edges = 0;
    std::unordered_map<uint32_t, uint32_t> node_pos;             // node id -> position in nodes
    std::unordered_map<uint32_t, std::vector<kc_edge>> in_edges;  // by input_id, in insertion order
    std::unordered_map<uint32_t, std::vector<kc_edge>> out_edges; // by output_id, in insertion order
};

struct NodeGraph {
    std::vector<Node> nodes;
    std::vector<kc_edge> edges;
    uint32_t node_id_counter = 0;
    uint64_t version = 0;      // bumped by every change to nodes / edges (touch())
    mutable GraphIndex idx;

    void touch() { ++version; }
    const GraphIndex &index() const;
    bool index_is_current() const;
    void appended_node(bool index_was_current);  // after nodes.push_back: bumps the version, patches a current index
    void appended_edge(bool index_was_current);  // after edges.push_back
    void erased_edge(const kc_edge &e, bool index_was_current);  // after edges.erase
    const std::vector<kc_edge> &edges_into(uint32_t id) const;
    const std::vector<kc_edge> &edges_out_of(uint32_t id) const;

    const Node *find(uint32_t id) const;
    Node *find(uint32_t id);
    uint32_t new_id();
    int add_node(Node n, uint32_t *id);
    int add_node_with_id(Node n);
    int connect(uint32_t on, uint32_t in, uint32_t os, uint32_t is);
    int try_connect(uint32_t on, uint32_t in, uint32_t os, uint32_t is);
    int remove_edge(kc_edge e);
    int remove_node(uint32_t id, std::vector<kc_edge> *removed);
    int disconnect_slot(uint32_t id, int side, uint32_t slot, std::vector<kc_edge> *removed);
    std::vector<uint32_t> get_children(uint32_t id) const;
    std::vector<uint32_t> get_children_recursive(uint32_t id) const;
    std::vector<uint32_t> get_parents(uint32_t id) const;
    std::vector<uint32_t> output_ids() const;
    int rename_output_node(uint32_t id, const std::string &new_name, std::string *old_name);
    std::vector<Slot> input_slots_of_graph() const;
    std::vector<Slot> output_slots_of_graph() const;
};

std::vector<Slot> node_input_slots(const Node &n, bool *unimplemented = nullptr);
std::vector<Slot> node_output_slots(const Node &n, bool *unimplemented = nullptr);
Node node_from_desc(const kc_node_desc &d);

int graph_from_json(const std::string &text, NodeGraph &g);
std::string graph_to_json(const NodeGraph &g);

}  // namespace kc

struct kc_node_graph {
    kc::NodeGraph g;
};

namespace kc {

struct SlotData {
    uint32_t node_id, slot_id;
    kc_image *image;  // retained
};

// A vector of plain records with room for N of them inside the object: the evaluator builds half a dozen
// two-or-three-element lists per node (inputs, resized, assigned, results ...) and the heap round trips for
// them were a measurable part of the ~0.7 us a node costs on the host.
template <class T, size_t N> class SmallVec {
    static_assert(std::is_trivially_copyable<T>::value, "SmallVec holds plain records only");
    T inl_[N];
    T *p_ = inl_;
    size_t n_ = 0, cap_ = N;

    void grow(size_t want)
    {
        size_t cap = cap_ * 2 > want ? cap_ * 2 : want;
        T *q = static_cast<T *>(::operator new(cap * sizeof(T)));
        std::memcpy(static_cast<void *>(q), p_, n_ * sizeof(T));
        if (p_ != inl_) ::operator delete(p_);
        p_ = q;
        cap_ = cap;
    }

public:
    SmallVec() = default;
    SmallVec(const T *first, size_t n) { assign(first, n); }
    SmallVec(const SmallVec &o) { assign(o.p_, o.n_); }
    SmallVec &operator=(const SmallVec &o)
    {
        if (this != &o) assign(o.p_, o.n_);
        return *this;
    }
    ~SmallVec()
    {
        if (p_ != inl_) ::operator delete(p_);
    }
    void assign(const T *first, size_t n)
    {
        n_ = 0;
        if (n > cap_) grow(n);
        if (n) std::memcpy(static_cast<void *>(p_), first, n * sizeof(T));
        n_ = n;
    }
    void push_back(const T &v)
    {
        if (n_ == cap_) {
            const T copy = v;  // v may live in this vector
            grow(n_ + 1);
            p_[n_++] = copy;
            return;
        }
        p_[n_++] = v;
    }
    void clear() { n_ = 0; }
    void erase_at(size_t i)
    {
        std::memmove(static_cast<void *>(p_ + i), p_ + i + 1, (n_ - i - 1) * sizeof(T));
        --n_;
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T *data() { return p_; }
    const T *data() const { return p_; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }
    T *begin() { return p_; }
    T *end() { return p_ + n_; }
    const T *begin() const { return p_; }
    const T *end() const { return p_ + n_; }
};
using SlotList = SmallVec<SlotData, 8>;
using EdgeList = SmallVec<kc_edge, 8>;

struct EmbeddedSlotData {
    uint32_t slot_data_id, slot_id;
    kc_image *image;
    // row-band evaluation only (bands.cpp): full_h != 0 => `image` holds logical rows band_y0 .. of a full_h-row image
    int32_t band_y0 = 0;
    uint32_t full_h = 0;
};

}  // namespace kc

struct kc_tex_pro {
    uint64_t memory_threshold = 0;
    std::recursive_mutex mu;
};

struct kc_live_graph {
    kc_tex_pro *tp = nullptr;
    kc::NodeGraph g;
    // by producing node, slots in the order the node returned them (a linear list made every look-up and every
    // replacement a scan of all cached slots: quadratic with use_cache on a long graph)
    std::unordered_map<uint32_t, kc::SmallVec<kc::SlotData, 4>> slot_datas;
    const kc::SmallVec<kc::SlotData, 4> &slots_of(uint32_t node) const;
    std::vector<kc::EmbeddedSlotData> embedded;
    std::vector<kc::SlotData> input_slot_datas;
    std::unordered_map<uint32_t, int> node_state;  // looked up a dozen times per node and evaluation
    std::unordered_set<uint32_t> changed;  // reported in ascending id order (kc_live_graph_changed_consume)
    bool auto_update = false;
    bool use_cache = false;
    uint64_t walks = 0;  // evaluations that were walked (not replayed): the first one builds plain chains, see await_clean
    std::string base_dir;
    int depth = 0;  // nesting depth of Graph nodes (recursion guard)
    struct ReplayEntry;
    ReplayEntry *replay = nullptr;  // the last evaluation that qualified, recorded for replay (replay.cpp)
    void replay_clear();

    ~kc_live_graph();
    void clear_data();
    void remove_nodes_data(uint32_t id);
    const kc::SlotData *find_slot(uint32_t node, uint32_t slot) const;
    int state_of(uint32_t id, int *st) const;
    int set_state(uint32_t id, int st);
    int force_state(uint32_t id, int st);
    void reset_node_states();
    int add_node(kc::Node n, uint32_t *id);
    int add_node_with_id(kc::Node n);
    int remove_node(uint32_t id);
    int connect(uint32_t on, uint32_t in, uint32_t os, uint32_t is);
    int remove_edge(kc_edge e);
    int disconnect_slot(uint32_t id, int side, uint32_t slot);
    int ensure_clean(uint32_t id);
    int import_slot_data(uint32_t node, uint32_t slot, kc_image *image);  // partition.cpp
    int await_clean(uint32_t id);
    int await_clean_walk(uint32_t id);
    int update();
    int process_one(uint32_t id);
};

// Multi-GPU placement plan (partition.cpp)
struct kc_partition {
    int world = 1, home = 0, n_levels = 1;
    int kind = KC_PLAN_BRANCHES;
    bool gather = true;  // KC_PLAN_BANDS: the bands end on the home rank (kc_partition_set_gather)
    uint32_t root = 0;
    std::vector<kc_placement> nodes;  // topological order
    std::vector<kc_transfer> xfers;   // execution order: level, producer's topological position, slot, destination
    std::vector<kc_band_range> bands;  // KC_PLAN_BANDS: rows [y0, y1) of the requested node per rank
    uint32_t full_w = 0, full_h = 0;
    double est_single = 0.0, est_branches = 0.0, est_bands = -1.0;  // what KC_PARTITION_AUTO compared (partition.cpp)
};

namespace kc {
// Row-band evaluation (bands.cpp)
int band_evaluate(kc_live_graph &lg, uint32_t root, uint32_t slot, int32_t y0, int32_t y1, kc_image **out);
int band_source_rows(kc_live_graph &lg, uint32_t root, int32_t y0, int32_t y1, std::vector<kc_band_rows> &out);
int partition_plan(kc_live_graph &lg, uint32_t root, int world, int policy, kc_partition **out);
int band_plan_info(kc_live_graph &lg, uint32_t root, kc_size *size, bool *rgba);  // the requested node's logical size, if the band walk takes the graph
// Replay of a recorded evaluation (replay.cpp)
struct ReplayRecorder;
int replay_try(kc_live_graph &lg, uint32_t id, bool *hit);
ReplayRecorder *replay_begin(kc_live_graph &lg, uint32_t id);
void replay_end(kc_live_graph &lg, uint32_t id, ReplayRecorder *r, int s);
// The exchange (comm.cpp): a shared-memory mailbox + IPC copies or RCCL.  Callers hold the context lock.
int comm_unique_id(void *id, size_t bytes);
int comm_init(int rank, int world, const void *id, size_t bytes);
int comm_destroy();
void comm_info(int *rank, int *world);
void comm_stats(uint64_t *planes_sent, uint64_t *planes_received, uint64_t *bytes_sent);
int comm_exchange(kc_live_graph &lg, const kc_transfer *t, uint32_t n);
int comm_evaluate_partitioned(kc_live_graph &lg, const kc_partition &plan, uint32_t root, kc_image **out);
int comm_gather_bands(kc_image *band, int32_t y0, uint32_t full_h, int home, kc_image **out);
const char *comm_wire_name();
void comm_blocks_freed();  // kc_pool_trim has given blocks back to the driver
void comm_sync();
// process_node, src/node/node_type.rs:213-248: inputs in edge insertion order.
int process_node(kc_live_graph &lg, const Node &node, const SlotList &inputs, const std::vector<kc_edge> &edges,
                 SlotList &out);
}  // namespace kc
