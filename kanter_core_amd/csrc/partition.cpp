// Multi-GPU placement of a graph evaluation: which rank (one process per GPU) evaluates which node, and
// which slots have to move between ranks.
//
// The reference has no distributed layer; what makes one possible is its readiness rule -- a node needs
// nothing but its parents' slot data (src/engine.rs:213-275) -- so independent branches of a graph can be
// evaluated anywhere and only the edges that cross a rank boundary carry data.  This file computes that cut,
// deterministically from the graph alone, so every rank (each holds the same NodeGraph) derives the same plan
// without talking to the others; the host above the C ABI moves the planes (RCCL send/recv over xGMI, see
// kanter_core_amd/multi_gpu.py and INTEGRATION.md) and hands them back with kc_live_graph_import_slot_data.
//
//   1. the ancestors of the requested node, in topological order (a cycle is an error);
//   2. node kinds: SOURCE (Embed / Image / Input*: holds data somebody embedded), REPLICATED (no source among
//      its ancestors -- Value nodes and what is built from them: constants, evaluated wherever needed, never
//      sent), COMPUTE (everything else);
//   3. components: a COMPUTE node continues its parent's component when it is that parent's only COMPUTE child
//      and has no other COMPUTE parent; fan-in and fan-out both start a new component.  (Sources do not count
//      as parents here: B feeding every other node of a chain must not cut the chain.)
//   4. placement: components downstream of a fan-in ("join region") go to the home rank, so that all branch
//      results converge on ONE rank over distinct xGMI links at the same time (a gather) instead of hopping
//      through a tree of ranks; the branch components are list-scheduled over the ranks by estimated finish
//      time.  KC_PARTITION_SPREAD ignores transfer cost in that estimate (use every GPU; the BASELINE multi-GPU
//      configs), KC_PARTITION_AUTO charges it -- and then keeps small graphs on one GPU, which is the right
//      answer more often than not: one 4096x4096 RGBA slot over a 153 GB/s link (1.3 ms) costs as much as a
//      dozen fused Mix chains of any length (105 us each at 5.7 TB/s);
//   5. sources live on the lowest rank that consumes them; every other edge from a non-replicated producer to a
//      consumer on another rank is a transfer.  One slot consumed on several ranks appears once per
//      destination, consecutively: a broadcast.
//
// The other way to use several GPUs is by ROWS (bands.cpp): every rank evaluates the whole graph for its band of the requested
// node -- pointwise nodes (src/node/mix.rs:136-192) need nothing from the other bands, a resize or HeightToNormal a few halo rows,
// computed redundantly -- and the only data that moves is the finished band, to the home rank's row offset (comm.cpp,
// comm_gather_bands): (world - 1) transfers of 1 / world of the result, each over its own xGMI link.  A plan of kind
// KC_PLAN_BANDS says which rows each rank takes.  KC_PARTITION_AUTO prices the three possibilities in the same unit (one fused
// RGBA Mix chain over the image = 40 B/px at the HBM rate; one RGBA slot over one link = 12 B/px at the link rate, 12 such units
// with the default rates) and takes the cheapest:
//     one GPU    sum of the node weights
//     branches   the list schedule's finish time, transfers charged (above)
//     bands      (sum of the node weights) / world  +  (transfer of the result) / world
// Config #4 (eight 16-node branches + add tree at 4096 x 4096): one GPU 5.1 units (one launch: 16 sources read once), bands
// 8.6 / 5.7 / 4.3 / 2.1 on 2 / 3 / 4 / 8 GPUs, branches 13-53: one GPU up to three ranks, bands from four on -- which is what the
// measured 0.555 ms of the single launch and 192 MB / ranks over a 153 GB/s link give.  A linear chain whose result must end on one GPU never pays (the result's transfer alone costs more than the
// chain): kc_partition_set_gather(plan, 0) leaves the bands where they are, for consumers that are row-parallel too.
#include <algorithm>

#include "kc_runtime.hpp"

using namespace kc;

namespace kc {

namespace {

enum Kind { SOURCE = 0, REPLICATED = 1, COMPUTE = 2 };

bool is_source(const Node &n) { return n.type == KC_NODE_EMBED || n.type == KC_NODE_IMAGE || n.is_input(); }

// relative cost of evaluating a node (one fused Mix chain ~ 1 whatever its length when nothing is cached)
double node_weight(const Node &n, bool use_cache)
{
    switch (n.type) {
    case KC_NODE_MIX: return use_cache ? 1.0 : 1.0 / 16.0;
    case KC_NODE_HEIGHT_TO_NORMAL: return 1.0;
    case KC_NODE_GRAPH: return 4.0;
    case KC_NODE_WRITE: return 4.0;
    default: return 0.0;  // aliasing nodes
    }
}

// one RGBA slot (three planes: Mix never reads the producer's alpha) over one xGMI link, in the units of node_weight:
// (12 B/px / link rate) / (40 B/px / HBM rate); 11.96 with the default 153 GB/s and 6.1 TB/s (kc_set_option "link_gbps" / "hbm_gbps")
double transfer_cost()
{
    const Context &c = ctx();
    return 0.3 * (double)c.hbm_gbps / (double)std::max(c.link_gbps, 1);
}

}  // namespace

int partition_plan(kc_live_graph &lg, uint32_t root, int world, int policy, kc_partition **out)
{
    const NodeGraph &g = lg.g;
    if (!g.find(root)) return KC_ERR_INVALID_NODE_ID;
    if (world < 1 || world > 1024 || (policy != KC_PARTITION_AUTO && policy != KC_PARTITION_SPREAD && policy != KC_PARTITION_BANDS)) {
        set_error("partition: world must be 1..1024 and policy KC_PARTITION_AUTO, KC_PARTITION_SPREAD or KC_PARTITION_BANDS");
        return KC_ERR_INVALID_ARG;
    }
    const double kTransferCost = transfer_cost();
    // ---- 1. ancestors in topological order (iterative post-order; an edge into the stack is a cycle)
    std::vector<uint32_t> topo;
    std::map<uint32_t, int> mark;  // 1 = on the stack, 2 = done
    {
        struct Frame {
            uint32_t id;
            std::vector<uint32_t> parents;
            size_t next;
        };
        std::vector<Frame> st;
        st.push_back(Frame{ root, g.get_parents(root), 0 });
        mark[root] = 1;
        while (!st.empty()) {
            Frame &f = st.back();
            if (f.next == f.parents.size()) {
                mark[f.id] = 2;
                topo.push_back(f.id);
                st.pop_back();
                continue;
            }
            const uint32_t p = f.parents[f.next++];
            if (!g.find(p)) continue;
            const int m = mark[p];
            if (m == 2) continue;
            if (m == 1) {
                set_error("graph has a cycle through node " + std::to_string(p));
                return KC_ERR_NODE_PROCESSING;
            }
            mark[p] = 1;
            st.push_back(Frame{ p, g.get_parents(p), 0 });
        }
    }
    const size_t n = topo.size();
    std::map<uint32_t, size_t> pos;
    for (size_t i = 0; i < n; ++i) pos[topo[i]] = i;
    auto needed = [&](uint32_t id) { return pos.count(id) != 0; };

    // ---- 2. kinds
    std::vector<int> kind(n, COMPUTE);
    for (size_t i = 0; i < n; ++i) {
        const Node &nd = *g.find(topo[i]);
        if (is_source(nd)) {
            kind[i] = SOURCE;
        } else if (nd.type == KC_NODE_GRAPH || nd.type == KC_NODE_WRITE) {
            kind[i] = COMPUTE;  // a nested graph may read files; a Write has a side effect: exactly one rank runs it
        } else {
            bool all_repl = true;
            for (uint32_t p : g.get_parents(topo[i]))
                if (needed(p)) all_repl &= kind[pos[p]] == REPLICATED;
            kind[i] = all_repl ? REPLICATED : COMPUTE;
        }
    }

    // ---- 3. components over the COMPUTE nodes
    std::vector<std::vector<size_t>> cparents(n), cchildren(n);
    for (size_t i = 0; i < n; ++i) {
        if (kind[i] != COMPUTE) continue;
        for (uint32_t p : g.get_parents(topo[i])) {
            if (!needed(p) || kind[pos[p]] != COMPUTE) continue;
            cparents[i].push_back(pos[p]);
            cchildren[pos[p]].push_back(i);
        }
    }
    std::vector<int> comp(n, -1);
    struct Comp {
        std::vector<size_t> members;
        std::set<int> inputs;
        double weight = 0.0;
        bool join = false;
        int rank = -1, level = 0;
        double finish = 0.0;
    };
    std::vector<Comp> comps;
    for (size_t i = 0; i < n; ++i) {
        if (kind[i] != COMPUTE) continue;
        if (cparents[i].size() == 1 && cchildren[cparents[i][0]].size() == 1) {
            comp[i] = comp[cparents[i][0]];
        } else {
            comp[i] = (int)comps.size();
            comps.emplace_back();
        }
        Comp &c = comps[comp[i]];
        c.members.push_back(i);
        c.weight += node_weight(*g.find(topo[i]), lg.use_cache);
        for (size_t p : cparents[i])
            if (comp[p] != comp[i]) c.inputs.insert(comp[p]);
    }
    // join region: fan-in components and everything downstream of one (components are numbered in
    // topological order, so one forward pass settles it)
    for (auto &c : comps) {
        c.join = c.inputs.size() >= 2;
        for (int i : c.inputs) c.join |= comps[i].join;
    }

    // ---- 4. placement
    const int home = 0;
    std::vector<double> avail((size_t)world, 0.0);
    // A branch result that a join consumes has to reach the home rank: KC_PARTITION_AUTO charges that hop to every
    // placement away from home (SPREAD does not look at transfer costs at all).
    std::vector<char> feeds_join(comps.size(), 0);
    for (auto &c : comps)
        if (c.join)
            for (int i : c.inputs) feeds_join[(size_t)i] = 1;
    // branch components first (they never depend on a join), then the join region: a join placed early would
    // make the home rank look busy until its remote inputs could have arrived and push every later branch away
    for (int pass = 0; pass < 2; ++pass)
        for (auto &c : comps) {
            if (c.join != (pass == 1)) continue;
            const double w = std::max(c.weight, 1.0 / 64.0);
            int best = home;
            if (!c.join) {
                double best_finish = 0.0;
                int best_local = -1;
                for (int r = 0; r < world; ++r) {
                    double ready = 0.0;
                    int local = 0;
                    for (int i : c.inputs) {
                        const bool same = comps[i].rank == r;
                        local += same;
                        ready = std::max(ready, comps[i].finish + ((same || policy == KC_PARTITION_SPREAD) ? 0.0 : kTransferCost));
                    }
                    double finish = std::max(avail[(size_t)r], ready) + w;
                    if (policy == KC_PARTITION_AUTO && feeds_join[(size_t)(&c - comps.data())] && r != home) finish += kTransferCost;
                    if (best_local < 0 || finish < best_finish - 1e-12 || (finish < best_finish + 1e-12 && local > best_local)) {
                        best = r;
                        best_finish = finish;
                        best_local = local;
                    }
                }
            }
            double ready = 0.0;
            for (int i : c.inputs) {
                ready = std::max(ready, comps[i].finish + (comps[i].rank == best ? 0.0 : kTransferCost));
                c.level = std::max(c.level, comps[i].level + (comps[i].rank == best ? 0 : 1));
            }
            c.rank = best;
            c.finish = std::max(avail[(size_t)best], ready) + w;
            avail[(size_t)best] = c.finish;
        }

    // ---- 5. sources, transfers
    std::vector<int> rank(n, -1);
    for (size_t i = 0; i < n; ++i)
        if (kind[i] == COMPUTE) rank[i] = comps[comp[i]].rank;
    auto consumers = [&](size_t i) {
        std::vector<size_t> c;
        for (uint32_t ch : g.get_children(topo[i]))
            if (needed(ch) && kind[pos[ch]] == COMPUTE) c.push_back(pos[ch]);
        return c;
    };
    for (size_t i = 0; i < n; ++i) {
        if (kind[i] != SOURCE) continue;
        int r = -1;
        for (size_t c : consumers(i)) r = r < 0 ? rank[c] : std::min(r, rank[c]);
        rank[i] = r < 0 ? home : r;  // a source nobody computes on is the requested node itself
    }
    // levels again, now that the sources have their ranks: a component that reads a source held by another rank runs after that
    // transfer (level 0), so what it sends is level 1 -- the transfers of one level never depend on each other (comm.cpp works
    // through a level's sends before its receives).  Components are numbered in topological order.
    for (auto &c : comps) {
        c.level = 0;
        for (int i : c.inputs) c.level = std::max(c.level, comps[i].level + (comps[i].rank == c.rank ? 0 : 1));
        for (size_t m : c.members)
            for (uint32_t p : g.get_parents(topo[m]))
                if (needed(p) && kind[pos[p]] == SOURCE && rank[pos[p]] != c.rank) c.level = std::max(c.level, 1);
    }
    kc_partition *P = new kc_partition();
    P->world = world;
    P->home = home;
    for (size_t i = 0; i < n; ++i)
        P->nodes.push_back(kc_placement{ topo[i], kind[i] == REPLICATED ? -1 : rank[i], kind[i] == COMPUTE ? comp[i] : -1, kind[i] });
    struct Key {
        int level;
        size_t ppos;
        uint32_t slot;
        int dst;
        bool operator<(const Key &o) const { return std::tie(level, ppos, slot, dst) < std::tie(o.level, o.ppos, o.slot, o.dst); }
    };
    std::set<Key> seen;
    int levels = 0;
    for (size_t i = 0; i < n; ++i) {
        if (kind[i] != COMPUTE) continue;
        for (auto &e : g.edges_into(topo[i])) {
            if (!needed(e.output_id)) continue;
            const size_t p = pos[e.output_id];
            if (kind[p] == REPLICATED || rank[p] == rank[i]) continue;
            const int level = kind[p] == COMPUTE ? comps[comp[p]].level : 0;
            seen.insert(Key{ level, p, e.output_slot, rank[i] });
            levels = std::max(levels, level + 1);
        }
    }
    for (auto &k : seen) P->xfers.push_back(kc_transfer{ topo[k.ppos], k.slot, rank[k.ppos], k.dst, k.level });
    P->n_levels = std::max(levels, 1);
    P->root = root;

    // ---- 6. one GPU, branches or row bands
    double single = 0.0, branches = 0.0;
    for (auto &c : comps) {
        single += std::max(c.weight, 1.0 / 64.0);
        branches = std::max(branches, c.finish);
    }
    // On ONE rank a graph that is pointwise from its sources to the requested node runs as one launch when nothing is cached
    // (joined chains, up to KC_CHAIN_MAX_IN planes per channel: csrc/runtime.cpp): it reads every source once and writes the
    // result once, whatever the number of nodes -- config #4's 135 nodes are 204 B/px, not the 304 the per-component sum gives.
    // In units of 40 B/px:
    if (!lg.use_cache) {
        uint32_t sources = 0;
        bool pointwise = true;
        for (size_t i = 0; i < n; ++i) {
            const Node &nd = *g.find(topo[i]);
            if (kind[i] == SOURCE) ++sources;
            else if (kind[i] == COMPUTE)
                pointwise &= nd.type == KC_NODE_MIX || nd.type == KC_NODE_SEPARATE_RGBA || nd.type == KC_NODE_COMBINE_RGBA || nd.is_output();
        }
        if (pointwise && sources >= 1 && sources <= (uint32_t)KC_CHAIN_MAX_IN) single = std::min(single, (3.0 * sources + 3.0) * 4.0 / 40.0);
    }
    bool all_home = P->xfers.empty();
    for (size_t i = 0; i < n; ++i) all_home &= kind[i] != COMPUTE || rank[i] == home;
    P->kind = all_home ? KC_PLAN_SINGLE : KC_PLAN_BRANCHES;
    P->est_single = single;
    P->est_branches = all_home ? single : branches;
    // bands need the requested node's height (every rank must cut the same rows) and a graph the band walk can take
    kc_size rsize{ 0, 0 };
    bool rrgba = true;
    std::string band_error;
    bool bands_ok = false;
    if (world > 1) {
        const std::string saved = last_error();
        bands_ok = band_plan_info(lg, root, &rsize, &rrgba) == KC_OK && rsize.height >= (uint32_t)world;
        if (!bands_ok) band_error = rsize.height && rsize.height < (uint32_t)world ? "fewer rows than ranks" : last_error();
        set_error(saved);
    }
    P->est_bands = bands_ok ? single / world + kTransferCost * (rrgba ? 1.0 : 1.0 / 3.0) / world : -1.0;
    if (policy == KC_PARTITION_BANDS && !bands_ok) {
        delete P;
        set_error("partition: no row-band plan for this graph (" + (world == 1 ? std::string("one rank") : band_error) +
                  "); sources must be embedded, whole or as constant placeholders of their size, before the plan is made");
        return KC_ERR_UNSUPPORTED;
    }
    if (bands_ok && (policy == KC_PARTITION_BANDS || (policy == KC_PARTITION_AUTO && P->est_bands < P->est_branches - 1e-9))) {
        P->kind = KC_PLAN_BANDS;
        P->xfers.clear();
        P->n_levels = 1;
        for (auto &pl : P->nodes) {
            pl.rank = -1;  // every rank: its rows
            pl.component = -1;
        }
        P->full_h = rsize.height;
        P->full_w = rsize.width;
        uint32_t start = 0;
        for (int r = 0; r < world; ++r) {
            const uint32_t rows = rsize.height / (uint32_t)world + ((uint32_t)r < rsize.height % (uint32_t)world ? 1u : 0u);
            P->bands.push_back(kc_band_range{ (int32_t)start, (int32_t)(start + rows) });
            start += rows;
        }
    }
    *out = P;
    return KC_OK;
}

}  // namespace kc

// Received slot data takes the place of evaluating the producer on this rank (engine.rs:34-57 stores a
// finished node's outputs the same way): the slot is replaced and the node is Clean.
int kc_live_graph::import_slot_data(uint32_t node, uint32_t slot, kc_image *image)
{
    if (!g.find(node)) return KC_ERR_INVALID_NODE_ID;
    auto &mine = slot_datas[node];
    for (size_t i = mine.size(); i-- > 0;)
        if (mine[i].slot_id == slot) {
            image_release(mine[i].image);
            mine.erase_at(i);
        }
    image_retain(image);
    mine.push_back(SlotData{ node, slot, image });
    // new data for this node: whatever was computed from the old one is out of date (the same propagation a
    // connect() triggers, src/live_graph.rs:515-537), the node itself is up to date
    for (uint32_t c : g.get_children(node)) KC_TRY(set_state(c, KC_STATE_DIRTY));
    node_state[node] = KC_STATE_CLEAN;
    changed.insert(node);
    return KC_OK;
}
