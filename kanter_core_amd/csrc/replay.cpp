// Replay of a recorded evaluation.
//
// The reference re-evaluates a graph by walking it node by node (src/engine.rs:200-307: readiness, process_node, slot
// bookkeeping per node); so does graph.cpp, at ~0.3 us of host time per node -- 15 us for the 32-node BASELINE graph, whatever
// the image size, against a kernel that takes 3 us at 256 x 256 and 9 us at 1024 x 1024 (the reference's own sizes,
// tests/integration_tests.rs:47-49).  An editor re-evaluating a graph after re-plugging the same input, or a batch job
// running one graph over and over, repeats EXACTLY the same walk: same nodes, same states, same slot data going in, same
// chain program coming out.  So await_clean records what an evaluation did when that was simple enough to describe --
//   * use_cache == false, no auto update, fusion on;
//   * every node it processed is a Mix / Value / CombineRgba / SeparateRgba / Output / Embed node (an Embed node is processed
//     again whenever its data has been dropped, src/engine.rs:58-75; what it yields is the embedded image, whose identity
//     is part of the check below);
//   * the device work was a sequence of plain chain launches (up to 32: config #4's eight branches and its add tree are nine),
//     each reading planes that existed before the evaluation or results of earlier launches of the sequence, the last results
//     being the requested node's planes (constants aside);
//   * afterwards no processed node but the requested one holds slot data --
// and the next await_clean of that node first checks whether the world looks exactly as it did before the recorded run:
//   * the same graph, by content (every node's type and parameters, every edge, in order: hashed each time, no reliance on
//     mutation hooks);
//   * the same state for every node;
//   * the same slot data on the same nodes and the same embedded images, by image identity (the recording holds a reference on
//     each, so an address cannot have been recycled); the requested node's own previous result is not an input and is ignored.
// If so the walk is skipped: fresh result planes for every launch, the recorded programs with the new output pointers (and the
// new pointers of earlier results where a launch reads one) through the same dispatch as ever (chain_dispatch: ahead-of-time
// kernel / specialised kernel / interpreter, cache policy), then the recorded bookkeeping -- states, dropped slot data, the changed set, the requested node's new slot.  Everything observable afterwards
// is what the walk would have left; the planes hold the same bits because they come from the same program on the same inputs.
// Anything else -- a different edge, a changed Mix type, another source image, a node that was Clean last time -- fails the
// check and takes the walk, which records again.
#include "kc_runtime.hpp"

using namespace kc;

struct kc_live_graph::ReplayEntry {
    uint32_t root = 0;
    uint64_t ghash = 0;
    std::vector<std::pair<uint32_t, int>> pre_state;  // every node
    struct Held {
        uint32_t node, slot;
        kc_image *img;  // retained
    };
    std::vector<Held> pre_slots;  // every slot datum present before the run, the requested node's own aside
    std::vector<EmbeddedSlotData> embedded;  // images retained
    std::vector<std::pair<uint32_t, int>> post_state;  // nodes whose state the run changed
    std::vector<uint32_t> dropped;                     // nodes whose slot data the run removed (the root included if it had any)
    std::vector<uint32_t> changed_ids;
    std::vector<ReplayLaunch> launches;
    uint32_t w = 0, h = 0;  // the requested node's image
    uint32_t root_slot = 0;
    int n_planes = 0;
    int out_l[4] = { -1, -1, -1, -1 };  // >= 0: a result of that launch ...
    int out_b[4] = { -1, -1, -1, -1 };  // ... channel out_b; out_l < 0: a constant plane
    float cval[4] = { 0, 0, 0, 0 };

    ~ReplayEntry()
    {
        for (auto &h : pre_slots) image_release(h.img);
        for (auto &em : embedded) image_release(em.image);
    }
};

namespace kc {

bool replay_enabled()
{
    static const bool on = !(std::getenv("KC_REPLAY") && std::atoi(std::getenv("KC_REPLAY")) == 0);
    return on && ctx().replay;
}

static uint64_t mix64(uint64_t h, uint64_t v)
{
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return h * 0xBF58476D1CE4E5B9ull;
}

// Everything about the graph an evaluation depends on (names and paths do not matter for the node types a recording allows).
static uint64_t graph_content_hash(const NodeGraph &g)
{
    uint64_t h = 0x243F6A8885A308D3ull;
    h = mix64(h, g.nodes.size());
    for (auto &n : g.nodes) {
        uint32_t vbits;
        std::memcpy(&vbits, &n.value, 4);
        h = mix64(h, ((uint64_t)n.node_id << 32) | (uint32_t)n.type);
        h = mix64(h, ((uint64_t)(uint32_t)n.mix_type << 32) | vbits);
        h = mix64(h, ((uint64_t)n.embed_id << 32) | (uint32_t)n.policy);
        h = mix64(h, ((uint64_t)n.policy_slot << 32) | (uint32_t)n.filter);
        h = mix64(h, ((uint64_t)n.policy_size.width << 32) | n.policy_size.height);
    }
    h = mix64(h, g.edges.size());
    for (auto &e : g.edges) {
        h = mix64(h, ((uint64_t)e.output_id << 32) | e.input_id);
        h = mix64(h, ((uint64_t)e.output_slot << 32) | e.input_slot);
    }
    return h;
}

static bool recordable_type(int t)
{
    return t == KC_NODE_MIX || t == KC_NODE_VALUE || t == KC_NODE_COMBINE_RGBA || t == KC_NODE_SEPARATE_RGBA ||
           t == KC_NODE_OUTPUT_GRAY || t == KC_NODE_OUTPUT_RGBA || t == KC_NODE_EMBED;
}

static size_t slot_count_except(const kc_live_graph &lg, uint32_t node)
{
    size_t n = 0;
    for (auto &kv : lg.slot_datas)
        if (kv.first != node) n += kv.second.size();
    return n;
}

// *hit = true: the evaluation has been replayed, `id` is Clean and holds its result.
int replay_try(kc_live_graph &lg, uint32_t id, bool *hit)
{
    *hit = false;
    const kc_live_graph::ReplayEntry *e = lg.replay;
    if (!e || e->root != id || lg.use_cache || lg.auto_update || !ctx().fusion || !replay_enabled()) return KC_OK;
    if (e->pre_state.size() != lg.node_state.size()) return KC_OK;
    for (auto &ps : e->pre_state) {
        auto it = lg.node_state.find(ps.first);
        if (it == lg.node_state.end() || it->second != ps.second) return KC_OK;
    }
    if (slot_count_except(lg, id) != e->pre_slots.size()) return KC_OK;
    for (auto &h : e->pre_slots) {
        const SlotData *sd = lg.find_slot(h.node, h.slot);
        if (!sd || sd->image != h.img) return KC_OK;
    }
    if (lg.embedded.size() != e->embedded.size()) return KC_OK;
    for (size_t i = 0; i < e->embedded.size(); ++i) {
        const EmbeddedSlotData &a = lg.embedded[i], &b = e->embedded[i];
        if (a.image != b.image || a.slot_data_id != b.slot_data_id || a.slot_id != b.slot_id || a.full_h != b.full_h || a.band_y0 != b.band_y0)
            return KC_OK;
    }
    if (graph_content_hash(lg.g) != e->ghash) return KC_OK;

    // ---- the launches ----
    Context &c = ctx();
    std::vector<kc_plane *> outs(e->launches.size() * KC_CHAIN_MAX_BATCH, nullptr);  // [launch][channel]
    auto drop_outs = [&] {
        for (auto *o : outs) plane_release(o);
    };
    for (size_t li = 0; li < e->launches.size(); ++li) {
        const ReplayLaunch &L = e->launches[li];
        ChainProgram P = L.prog;
        for (int b = 0; b < L.batch; ++b) {
            kc_plane *&o = outs[li * KC_CHAIN_MAX_BATCH + b];
            int s = plane_new_mem(L.w, L.h, &o);
            if (s != KC_OK) {
                drop_outs();
                return s;
            }
            P.out[b] = o->dptr;
            P.out_pitch[b] = (uint32_t)(o->pitch / 16);
            for (uint32_t k = 0; k < P.n_in; ++k)
                if (L.in_from[b][k] >= 0) {  // a result of an earlier launch of this replay
                    const kc_plane *src = outs[(size_t)L.in_from[b][k] * KC_CHAIN_MAX_BATCH + L.in_ch[b][k]];
                    P.in[b][k] = src->dptr;
                    P.in_pitch[b][k] = (uint32_t)(src->pitch / 16);
                }
        }
        const uint64_t px4 = 4ull * L.w * L.h;
        P.nt_mask = chain_cache_policy(L.in_refs, P.n_in, px4 * L.batch, px4 * L.batch);
        hipError_t he = chain_dispatch(P, L.batch, L.mode, L.w, L.h, outs[li * KC_CHAIN_MAX_BATCH]->pitch);
        if (he == hipErrorNotReady) {  // a program that joins two chains, and its kernel is not to be had (kc_set_specialize(0)): walk
            drop_outs();
            return KC_OK;
        }
        if (he != hipSuccess) {
            drop_outs();
            return hip_fail(he, "launch_chain (replay)");
        }
        c.launches++;
        c.alg_bytes += (uint64_t)L.batch * px4 * (P.n_in + 1);
    }
    c.counters["replayed_evaluations"]++;

    // ---- the result image ----
    kc_plane *planes[4] = { nullptr, nullptr, nullptr, nullptr };
    std::vector<kc_plane *> consts;
    for (int p = 0; p < e->n_planes; ++p) {
        if (e->out_l[p] >= 0) planes[p] = outs[(size_t)e->out_l[p] * KC_CHAIN_MAX_BATCH + e->out_b[p]];
        else {
            planes[p] = plane_new_const(e->w, e->h, e->cval[p]);
            consts.push_back(planes[p]);
        }
    }
    kc_image *img = image_new(e->n_planes, planes);  // retains the planes
    for (auto *q : consts) plane_release(q);
    drop_outs();  // intermediates go back to the pool (stream order: the launches that read them are enqueued)

    // ---- the bookkeeping the walk would have done ----
    for (uint32_t d : e->dropped) lg.remove_nodes_data(d);
    lg.slot_datas[id].push_back(SlotData{ id, e->root_slot, img });  // takes over the reference image_new returned
    for (auto &ps : e->post_state) lg.node_state[ps.first] = ps.second;
    for (uint32_t ch : e->changed_ids) lg.changed.insert(ch);
    *hit = true;
    return KC_OK;
}

// Snapshot before a walk; nullptr = this evaluation cannot be recorded.
struct ReplayRecorder {
    kc_live_graph::ReplayEntry *entry = nullptr;
    ReplayCapture cap;
    uint64_t launches0 = 0;
    std::unordered_set<uint32_t> changed0;
    // The planes of the held images (slot data, embedded images) that were RESIDENT when the recording began, by address: the
    // only planes a recorded launch may read from outside the sequence.  The entry's reference on the image keeps exactly those
    // blocks alive.  An embedded image that is still an unevaluated chain or a deferred resize (mix_process / as_type / resize
    // results can be embedded) reads the operands of ITS chain, which only its links hold: forced from outside between two
    // evaluations they go back to the pool and a replay would launch on recycled addresses.
    std::set<const void *> resident0;
};

static void note_resident(std::set<const void *> &set, const kc_image *img)
{
    for (int p = 0; p < img->n; ++p)
        if (img->planes[p]->kind == kc_plane::MEM) set.insert(img->planes[p]->dptr);
}

ReplayRecorder *replay_begin(kc_live_graph &lg, uint32_t id)
{
    Context &c = ctx();
    if (lg.use_cache || lg.auto_update || !c.fusion || !replay_enabled() || c.capture || lg.depth > 0) return nullptr;
    if (lg.g.nodes.size() > 4096) return nullptr;  // the check is linear in the graph; keep it negligible
    auto *r = new ReplayRecorder();
    auto *e = new kc_live_graph::ReplayEntry();
    r->entry = e;
    e->root = id;
    e->ghash = graph_content_hash(lg.g);
    e->pre_state.assign(lg.node_state.begin(), lg.node_state.end());
    for (auto &kv : lg.slot_datas)
        for (auto &sd : kv.second) {
            if (sd.node_id == id) continue;  // the requested node's previous result: replaced, never read
            image_retain(sd.image);
            e->pre_slots.push_back({ sd.node_id, sd.slot_id, sd.image });
            note_resident(r->resident0, sd.image);
        }
    for (auto &em : lg.embedded) {
        if (em.full_h != 0) {  // row-band sources: another evaluation path
            delete e;
            delete r;
            return nullptr;
        }
        image_retain(em.image);
        e->embedded.push_back(em);
        note_resident(r->resident0, em.image);
    }
    r->launches0 = c.launches;
    r->changed0 = lg.changed;
    c.capture = &r->cap;
    return r;
}

// After the walk (status `s`): keeps the recording if the evaluation qualified, forgets it otherwise.
void replay_end(kc_live_graph &lg, uint32_t id, ReplayRecorder *r, int s)
{
    if (!r) return;
    Context &c = ctx();
    c.capture = nullptr;
    std::unique_ptr<ReplayRecorder> own(r);
    std::unique_ptr<kc_live_graph::ReplayEntry> e(r->entry);
    lg.replay_clear();
    if (s != KC_OK || !r->cap.ok || r->cap.launches.empty() || c.launches - r->launches0 != r->cap.launches.size()) return;
    // The nodes the run processed are those whose state it changed and, possibly, those that held no data before it (a
    // Clean parent whose data had been dropped is made Dirty, processed and dropped again: same state before and after).
    // All of them must be of a recordable type.
    std::unordered_map<uint32_t, int> pre(e->pre_state.begin(), e->pre_state.end());
    if (pre.size() != lg.node_state.size()) return;
    {
        std::unordered_set<uint32_t> had;
        for (auto &h : e->pre_slots) had.insert(h.node);
        for (auto &n : lg.g.nodes)
            if (!had.count(n.node_id) && !recordable_type(n.type)) return;
    }
    for (auto &kv : lg.node_state) {
        auto it = pre.find(kv.first);
        if (it == pre.end()) return;
        if (it->second != kv.second) {
            const Node *n = lg.g.find(kv.first);
            if (!n || !recordable_type(n->type)) return;
            e->post_state.push_back({ kv.first, kv.second });
        }
    }
    // Nodes that were re-dirtied and brought back (a parent whose data had been dropped) end in the state they started in:
    // they were processed all the same, which shows in their slot data (dropped again) -- covered by the slot comparison below.
    // slot data afterwards: what was there before (same images), minus what the run dropped, plus the root's
    std::unordered_map<uint64_t, kc_image *> before;
    for (auto &h : e->pre_slots) before[((uint64_t)h.node << 32) | h.slot] = h.img;
    std::unordered_set<uint32_t> has_after;
    const SlotData *root_sd = nullptr;
    size_t root_slots = 0;
    for (auto &kv : lg.slot_datas)
        for (auto &sd : kv.second) {
            has_after.insert(sd.node_id);
            if (sd.node_id == id) {
                root_sd = &sd;
                ++root_slots;
                continue;
            }
            auto it = before.find(((uint64_t)sd.node_id << 32) | sd.slot_id);
            if (it == before.end() || it->second != sd.image) return;  // a processed node other than the root kept (new) data
        }
    if (!root_sd || root_slots != 1) return;
    std::unordered_set<uint32_t> dropped;
    for (auto &h : e->pre_slots)
        if (!has_after.count(h.node)) dropped.insert(h.node);
    dropped.insert(id);  // whatever the requested node held is replaced
    // a node that kept some slots but lost others cannot be described by "drop the node's data"
    for (auto &h : e->pre_slots)
        if (!dropped.count(h.node) && !lg.find_slot(h.node, h.slot)) return;
    e->dropped.assign(dropped.begin(), dropped.end());
    // the root's image: results of the recorded launches, or constants
    const kc_image *img = root_sd->image;
    ReplayCapture &cap = r->cap;
    e->w = img->w();
    e->h = img->h();
    e->root_slot = root_sd->slot_id;
    e->n_planes = img->n;
    for (int p = 0; p < img->n; ++p) {
        const kc_plane *pl = img->planes[p];
        e->out_l[p] = e->out_b[p] = -1;
        if (pl->kind == kc_plane::CONST) {
            e->cval[p] = pl->cval;
            continue;
        }
        for (size_t li = 0; li < cap.launches.size(); ++li)
            for (int b = 0; b < cap.launches[li].batch; ++b)
                if (cap.launches[li].planes[b] == pl && cap.launches[li].w == img->w() && cap.launches[li].h == img->h()) {
                    e->out_l[p] = (int)li;
                    e->out_b[p] = b;
                }
        if (e->out_l[p] < 0) return;  // a plane from somewhere else (a pass-through of a source ...)
    }
    // which inputs are results of earlier launches of the sequence: by address, the LATEST earlier launch that wrote there
    // (a pool block can be handed out twice during one evaluation; what a launch reads is what was written last)
    for (size_t li = 0; li < cap.launches.size(); ++li) {
        ReplayLaunch &L = cap.launches[li];
        for (int b = 0; b < L.batch; ++b)
            for (uint32_t k = 0; k < L.prog.n_in; ++k) {
                L.in_from[b][k] = L.in_ch[b][k] = -1;
                for (size_t lj = li; lj-- > 0 && L.in_from[b][k] < 0;)
                    for (int bj = 0; bj < cap.launches[lj].batch; ++bj)
                        if (cap.launches[lj].prog.out[bj] == L.prog.in[b][k]) {
                            if (cap.launches[lj].w != L.w || cap.launches[lj].h != L.h) return;  // (cannot be: chains are pointwise)
                            L.in_from[b][k] = (int)lj;
                            L.in_ch[b][k] = bj;
                        }
                // anything else must be a plane the entry itself keeps alive (see ReplayRecorder::resident0)
                if (L.in_from[b][k] < 0 && !r->resident0.count(L.prog.in[b][k])) return;
            }
    }
    for (uint32_t ch : lg.changed)
        if (!r->changed0.count(ch)) e->changed_ids.push_back(ch);
    // ids already in the set before the run stay in it either way
    e->launches = std::move(cap.launches);
    lg.replay = e.release();
}

}  // namespace kc

void kc_live_graph::replay_clear()
{
    delete replay;
    replay = nullptr;
}
