// Host graph runtime: NodeGraph (src/node_graph.rs), the node slot tables and process_node
// (src/node/node_type.rs:140-267), LiveGraph state + result store (src/live_graph.rs) and the
// stream-ordered evaluator that replaces engine::process_loop (src/engine.rs:25-312): instead of
// a 1 ms scheduler tick spawning one OS thread per ready node, await_clean walks the dirty
// ancestors of the requested node in topological order on the calling thread and enqueues their
// kernels on the context's HIP stream; with use_cache == false intermediates are dropped as soon
// as all their children are done (src/engine.rs:58-75), which is also what lets pointwise Mix
// chains stay lazy and run as one fused kernel.
#include <algorithm>
#include <cstring>

#include "kc_runtime.hpp"

namespace kc {

// ------------------------------------------------------------------------------------------
// Slot tables: Node::input_slots / output_slots, src/node/node_type.rs:140-211
// ------------------------------------------------------------------------------------------
std::vector<Slot> NodeGraph::input_slots_of_graph() const
{
    // src/node_graph.rs:285-298: one slot per Input node, SlotId = the inner NodeId
    std::vector<Slot> out;
    for (auto &n : nodes)
        if (n.is_input()) out.push_back({ n.text, n.node_id, n.type == KC_NODE_INPUT_GRAY ? SLOT_GRAY : SLOT_RGBA });
    return out;
}

std::vector<Slot> NodeGraph::output_slots_of_graph() const
{
    // src/node_graph.rs:300-313
    std::vector<Slot> out;
    for (auto &n : nodes)
        if (n.is_output()) out.push_back({ n.text, n.node_id, n.type == KC_NODE_OUTPUT_GRAY ? SLOT_GRAY : SLOT_RGBA });
    return out;
}

std::vector<Slot> node_input_slots(const Node &n, bool *unimplemented)
{
    if (unimplemented) *unimplemented = false;
    switch (n.type) {
    case KC_NODE_OUTPUT_GRAY: return { { "input", 0, SLOT_GRAY } };
    case KC_NODE_OUTPUT_RGBA: return { { "input", 0, SLOT_RGBA } };
    case KC_NODE_GRAPH: return n.graph ? n.graph->input_slots_of_graph() : std::vector<Slot>{};
    case KC_NODE_WRITE:
        // unimplemented!() in the reference (node_type.rs:154); Write takes one image in practice
        if (unimplemented) *unimplemented = true;
        return { { "input", 0, SLOT_GRAY_OR_RGBA } };
    case KC_NODE_MIX: return { { "left", 0, SLOT_GRAY_OR_RGBA }, { "right", 1, SLOT_GRAY_OR_RGBA } };
    case KC_NODE_HEIGHT_TO_NORMAL: return { { "input", 0, SLOT_GRAY } };
    case KC_NODE_SEPARATE_RGBA: return { { "input", 0, SLOT_RGBA } };
    case KC_NODE_COMBINE_RGBA:
        return { { "red", 0, SLOT_GRAY }, { "green", 1, SLOT_GRAY }, { "blue", 2, SLOT_GRAY }, { "alpha", 3, SLOT_GRAY } };
    default: return {};
    }
}

std::vector<Slot> node_output_slots(const Node &n, bool *unimplemented)
{
    if (unimplemented) *unimplemented = false;
    switch (n.type) {
    case KC_NODE_INPUT_GRAY: return { { "output", 0, SLOT_GRAY } };
    case KC_NODE_INPUT_RGBA: return { { "output", 0, SLOT_RGBA } };
    case KC_NODE_OUTPUT_GRAY: case KC_NODE_OUTPUT_RGBA: return {};
    case KC_NODE_GRAPH: return n.graph ? n.graph->output_slots_of_graph() : std::vector<Slot>{};
    case KC_NODE_IMAGE: case KC_NODE_EMBED: return { { "output", 0, SLOT_RGBA } };
    case KC_NODE_WRITE:
        if (unimplemented) *unimplemented = true;
        return {};
    case KC_NODE_VALUE: return { { "output", 0, SLOT_GRAY } };
    case KC_NODE_MIX: return { { "output", 0, SLOT_GRAY_OR_RGBA } };
    case KC_NODE_HEIGHT_TO_NORMAL: return { { "output", 0, SLOT_RGBA } };
    case KC_NODE_SEPARATE_RGBA:
        return { { "red", 0, SLOT_GRAY }, { "green", 1, SLOT_GRAY }, { "blue", 2, SLOT_GRAY }, { "alpha", 3, SLOT_GRAY } };
    case KC_NODE_COMBINE_RGBA: return { { "output", 0, SLOT_RGBA } };
    default: return {};
    }
}

// node_output_slots(n).size() without building the named slot list (asked once per evaluated node)
static size_t node_output_slot_count(const Node &n)
{
    switch (n.type) {
    case KC_NODE_OUTPUT_GRAY: case KC_NODE_OUTPUT_RGBA: case KC_NODE_WRITE: return 0;
    case KC_NODE_SEPARATE_RGBA: return 4;
    case KC_NODE_GRAPH: return node_output_slots(n).size();
    case KC_NODE_INPUT_GRAY: case KC_NODE_INPUT_RGBA: case KC_NODE_IMAGE: case KC_NODE_EMBED: case KC_NODE_VALUE:
    case KC_NODE_MIX: case KC_NODE_HEIGHT_TO_NORMAL: case KC_NODE_COMBINE_RGBA: return 1;
    default: return 0;
    }
}

// SlotType::fits, src/node/mod.rs:209-221
static bool slot_fits(int self, int other)
{
    if (self == SLOT_GRAY) return other == SLOT_GRAY || other == SLOT_GRAY_OR_RGBA;
    if (self == SLOT_RGBA) return other == SLOT_RGBA || other == SLOT_GRAY_OR_RGBA;
    return true;
}

Node node_from_desc(const kc_node_desc &d)
{
    Node n;
    n.node_id = d.node_id;
    n.type = d.node_type;
    n.mix_type = d.mix_type;
    n.value = d.value;
    n.embed_id = d.embed_id;
    if (d.text) n.text = d.text;
    if (d.graph) n.graph = std::make_shared<NodeGraph>(d.graph->g);
    n.policy = d.resize_policy;
    n.policy_slot = d.policy_slot;
    n.policy_size = d.policy_size;
    n.filter = d.resize_filter;
    return n;
}

// ------------------------------------------------------------------------------------------
// NodeGraph, src/node_graph.rs
// ------------------------------------------------------------------------------------------
const GraphIndex &NodeGraph::index() const
{
    // sizes are compared as well: a change that forgot touch() still cannot leave a stale index behind
    // unless it kept both counts
    if (idx.version != version || idx.n_nodes != nodes.size() || idx.n_edges != edges.size()) {
        idx.node_pos.clear();
        idx.in_edges.clear();
        idx.out_edges.clear();
        for (size_t i = 0; i < nodes.size(); ++i) idx.node_pos.emplace(nodes[i].node_id, (uint32_t)i);  // first wins, as the scan did
        for (auto &e : edges) {
            idx.in_edges[e.input_id].push_back(e);
            idx.out_edges[e.output_id].push_back(e);
        }
        idx.version = version;
        idx.n_nodes = nodes.size();
        idx.n_edges = edges.size();
    }
    return idx;
}

// Appending a node or an edge to a graph whose index is current patches the index instead of invalidating it: building
// an n-node chain through add_node / connect used to rebuild the whole index n times (a 20 000-node chain took two minutes).
bool NodeGraph::index_is_current() const
{
    return idx.version == version && idx.n_nodes == nodes.size() && idx.n_edges == edges.size();
}

void NodeGraph::appended_node(bool index_was_current)
{
    touch();
    if (!index_was_current) return;
    idx.node_pos.emplace(nodes.back().node_id, (uint32_t)(nodes.size() - 1));  // first wins, as in index()
    idx.version = version;
    idx.n_nodes = nodes.size();
}

// Removing an edge from a graph whose index is current patches the index too (re-plugging an input of a 32-node graph used to
// cost a rebuild of three hash tables of vectors: most of a connect()).  Call after edges.erase with the erased edge.
void NodeGraph::erased_edge(const kc_edge &e, bool index_was_current)
{
    touch();
    if (!index_was_current) return;
    auto drop = [&](std::vector<kc_edge> &v) {
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].output_id == e.output_id && v[i].input_id == e.input_id && v[i].output_slot == e.output_slot && v[i].input_slot == e.input_slot) {
                v.erase(v.begin() + (long)i);
                return;
            }
    };
    auto a = idx.in_edges.find(e.input_id);
    auto b = idx.out_edges.find(e.output_id);
    if (a == idx.in_edges.end() || b == idx.out_edges.end()) return;  // cannot be: leave the index stale, it is rebuilt on use
    drop(a->second);
    drop(b->second);
    idx.version = version;
    idx.n_edges = edges.size();
}

void NodeGraph::appended_edge(bool index_was_current)
{
    touch();
    if (!index_was_current) return;
    const kc_edge &e = edges.back();
    idx.in_edges[e.input_id].push_back(e);
    idx.out_edges[e.output_id].push_back(e);
    idx.version = version;
    idx.n_edges = edges.size();
}

static const std::vector<kc_edge> kNoEdges;

const std::vector<kc_edge> &NodeGraph::edges_into(uint32_t id) const
{
    const GraphIndex &ix = index();
    auto it = ix.in_edges.find(id);
    return it == ix.in_edges.end() ? kNoEdges : it->second;
}

const std::vector<kc_edge> &NodeGraph::edges_out_of(uint32_t id) const
{
    const GraphIndex &ix = index();
    auto it = ix.out_edges.find(id);
    return it == ix.out_edges.end() ? kNoEdges : it->second;
}

const Node *NodeGraph::find(uint32_t id) const
{
    const GraphIndex &ix = index();
    auto it = ix.node_pos.find(id);
    return it == ix.node_pos.end() ? nullptr : &nodes[it->second];
}

Node *NodeGraph::find(uint32_t id)
{
    const GraphIndex &ix = index();
    auto it = ix.node_pos.find(id);
    return it == ix.node_pos.end() ? nullptr : &nodes[it->second];
}

uint32_t NodeGraph::new_id()
{
    // :86-96
    uint32_t out = node_id_counter++;
    while (find(out)) out = node_id_counter++;
    return out;
}

// avoid_name_collision, :141-164
static std::string avoid_name_collision(const std::vector<std::string> &names, const std::string &name)
{
    std::string edit = name;
    auto taken = [&](const std::string &s) { return std::find(names.begin(), names.end(), s) != names.end(); };
    while (taken(edit)) {
        size_t us = edit.rfind('_');
        if (us != std::string::npos) {
            std::string head = edit.substr(0, us), num = edit.substr(us + 1);
            bool numeric = std::all_of(num.begin(), num.end(), [](unsigned char c) { return std::isdigit(c); });
            if (numeric) {
                uint32_t v = 0;
                bool parsed = !num.empty();
                uint64_t acc = 0;
                for (char c : num) {
                    acc = acc * 10 + (uint64_t)(c - '0');
                    if (acc > 0xFFFFFFFFull) parsed = false;
                }
                v = parsed ? (uint32_t)acc + 1u : 0u;  // wrapping_add(1), or 0 when the parse fails
                edit = head + "_" + std::to_string(v);
            } else {
                edit = head + "_0";
            }
        } else {
            edit = edit + "_0";
        }
    }
    return edit;
}

static int add_node_internal(NodeGraph &g, Node n, uint32_t id)
{
    // :166-189
    if (n.is_input() || n.is_output()) {
        if (n.text.empty()) n.text = "untitled";
        std::vector<std::string> names;
        for (auto &o : g.nodes)
            if (n.is_input() ? o.is_input() : o.is_output()) names.push_back(o.text);
        n.text = avoid_name_collision(names, n.text);
    }
    n.node_id = id;
    const bool cur = g.index_is_current();
    g.nodes.push_back(std::move(n));
    g.appended_node(cur);
    return KC_OK;
}

int NodeGraph::add_node(Node n, uint32_t *id)
{
    if (n.type < KC_NODE_INPUT_GRAY || n.type > KC_NODE_COMBINE_RGBA) {
        set_error("invalid NodeType");
        return KC_ERR_INVALID_NODE_TYPE;
    }
    uint32_t nid = new_id();
    KC_TRY(add_node_internal(*this, std::move(n), nid));
    if (id) *id = nid;
    return KC_OK;
}

int NodeGraph::add_node_with_id(Node n)
{
    if (n.type < KC_NODE_INPUT_GRAY || n.type > KC_NODE_COMBINE_RGBA) return KC_ERR_INVALID_NODE_TYPE;
    if (find(n.node_id)) return KC_ERR_INVALID_NODE_ID;  // :322-331
    uint32_t id = n.node_id;
    return add_node_internal(*this, std::move(n), id);
}

static int slot_type_of(const std::vector<Slot> &slots, uint32_t id, int *type)
{
    for (auto &s : slots)
        if (s.slot_id == id) {
            *type = s.slot_type;
            return KC_OK;
        }
    return KC_ERR_INVALID_SLOT_ID;
}

int NodeGraph::try_connect(uint32_t on, uint32_t in, uint32_t os, uint32_t is)
{
    // can_connect + push, :376-413 (no slot-type check here in the reference either)
    const Node *o = find(on), *i = find(in);
    if (!o || !i) return KC_ERR_INVALID_NODE_ID;
    int t;
    KC_TRY(slot_type_of(node_output_slots(*o), os, &t));
    KC_TRY(slot_type_of(node_input_slots(*i), is, &t));
    for (auto &e : edges_into(in))
        if (e.input_slot == is) return KC_ERR_SLOT_OCCUPIED;
    const bool cur = index_is_current();
    edges.push_back(kc_edge{ on, in, os, is });
    appended_edge(cur);
    return KC_OK;
}

int NodeGraph::connect(uint32_t on, uint32_t in, uint32_t os, uint32_t is)
{
    // :416-446
    const Node *o = find(on), *i = find(in);
    if (!o || !i) return KC_ERR_INVALID_NODE_ID;
    // Re-plugging the edge that is already the newest one (an editor re-connecting the same cable, bench.py's step): the
    // reference removes it and appends it again, which leaves `edges` exactly as it is -- nothing to do, and above all no
    // reason to throw the index away.  (Its slots were checked when it was first connected.)
    if (!edges.empty()) {
        const kc_edge &l = edges.back();
        if (l.output_id == on && l.input_id == in && l.output_slot == os && l.input_slot == is) return KC_OK;
    }
    int ot, it;
    KC_TRY(slot_type_of(node_output_slots(*o), os, &ot));
    KC_TRY(slot_type_of(node_input_slots(*i), is, &it));
    if (!slot_fits(ot, it)) return KC_ERR_INVALID_SLOT_TYPE;
    (void)disconnect_slot(in, KC_SIDE_INPUT, is, nullptr);
    for (auto &e : edges_into(in))
        if (e.output_id == on && e.output_slot == os && e.input_slot == is) return KC_ERR_INVALID_EDGE;
    const bool cur = index_is_current();
    edges.push_back(kc_edge{ on, in, os, is });
    appended_edge(cur);
    return KC_OK;
}

int NodeGraph::remove_edge(kc_edge e)
{
    for (size_t i = 0; i < edges.size(); ++i) {
        const kc_edge &c = edges[i];
        if (c.output_id == e.output_id && c.input_id == e.input_id && c.output_slot == e.output_slot && c.input_slot == e.input_slot) {
            if (!find(e.input_id)) return KC_ERR_INVALID_NODE_ID;
            const bool cur = index_is_current();
            const kc_edge gone = c;
            edges.erase(edges.begin() + (long)i);
            erased_edge(gone, cur);
            return KC_OK;
        }
    }
    return KC_ERR_INVALID_EDGE;
}

int NodeGraph::remove_node(uint32_t id, std::vector<kc_edge> *removed)
{
    // :473-493
    if (!find(id)) return KC_ERR_INVALID_NODE_ID;
    for (size_t i = edges.size(); i-- > 0;)
        if (edges[i].output_id == id || edges[i].input_id == id) {
            if (removed) removed->push_back(edges[i]);
            edges.erase(edges.begin() + (long)i);
            touch();
        }
    for (size_t i = 0; i < nodes.size(); ++i)
        if (nodes[i].node_id == id) {
            nodes.erase(nodes.begin() + (long)i);
            touch();
            break;
        }
    return KC_OK;
}

int NodeGraph::disconnect_slot(uint32_t id, int side, uint32_t slot, std::vector<kc_edge> *removed)
{
    // :496-515 (every edge on the slot goes)
    if (!find(id)) return KC_ERR_INVALID_NODE_ID;
    {
        // nothing on the slot (the usual case when a graph is being built): the index knows without a scan of every edge
        bool occupied = false;
        for (auto &e : side == KC_SIDE_INPUT ? edges_into(id) : edges_out_of(id))
            occupied = occupied || (side == KC_SIDE_INPUT ? e.input_slot : e.output_slot) == slot;
        if (!occupied) return KC_ERR_SLOT_NOT_OCCUPIED;
    }
    bool any = false;
    for (size_t i = edges.size(); i-- > 0;) {
        const kc_edge &e = edges[i];
        const bool hit = side == KC_SIDE_INPUT ? (e.input_id == id && e.input_slot == slot)
                                               : (e.output_id == id && e.output_slot == slot);
        if (hit) {
            if (removed) removed->insert(removed->begin(), e);
            const bool cur = index_is_current();
            const kc_edge gone = e;
            edges.erase(edges.begin() + (long)i);
            erased_edge(gone, cur);
            any = true;
        }
    }
    return any ? KC_OK : KC_ERR_SLOT_NOT_OCCUPIED;
}

std::vector<uint32_t> NodeGraph::get_children(uint32_t id) const
{
    std::vector<uint32_t> c;
    for (auto &e : edges_out_of(id)) c.push_back(e.input_id);
    std::sort(c.begin(), c.end());
    c.erase(std::unique(c.begin(), c.end()), c.end());
    return c;
}

std::vector<uint32_t> NodeGraph::get_children_recursive(uint32_t id) const
{
    // node_graph.rs:566-575 recurses child by child (and never returns on a cycle, which connect() does not
    // forbid); every caller only needs the set of descendants, so this is a worklist with a visited set.
    std::vector<uint32_t> out, work = get_children(id);
    std::set<uint32_t> seen(work.begin(), work.end());
    while (!work.empty()) {
        const uint32_t c = work.back();
        work.pop_back();
        out.push_back(c);
        for (uint32_t g : get_children(c))
            if (seen.insert(g).second) work.push_back(g);
    }
    return out;
}

std::vector<uint32_t> NodeGraph::get_parents(uint32_t id) const
{
    std::vector<uint32_t> p;
    for (auto &e : edges_into(id)) p.push_back(e.output_id);
    std::sort(p.begin(), p.end());
    p.erase(std::unique(p.begin(), p.end()), p.end());
    return p;
}

int NodeGraph::rename_output_node(uint32_t id, const std::string &new_name, std::string *old_name)
{
    // :232-269: the node's own name is taken out of the collision list first
    Node *n = find(id);
    if (!n) return KC_ERR_INVALID_NODE_ID;
    if (!n->is_output()) return KC_ERR_INVALID_NODE_TYPE;
    std::vector<std::string> names;
    bool skipped = false;
    for (auto &o : nodes) {
        if (!o.is_output()) continue;
        if (!skipped && o.text == n->text) {
            skipped = true;
            continue;
        }
        names.push_back(o.text);
    }
    if (old_name) *old_name = n->text;
    n->text = avoid_name_collision(names, new_name);
    return KC_OK;
}

std::vector<uint32_t> NodeGraph::output_ids() const
{
    std::vector<uint32_t> out;
    for (auto &n : nodes)
        if (n.is_output()) out.push_back(n.node_id);
    return out;
}

// ------------------------------------------------------------------------------------------
// process_node, src/node/node_type.rs:213-267 (+ resize_buffers, src/shared.rs:141-216)
// ------------------------------------------------------------------------------------------
static void release_all(SlotList &v)
{
    for (auto &sd : v) image_release(sd.image);
    v.clear();
}

static std::string resolve_path(const kc_live_graph &lg, const std::string &p)
{
    if (p.empty() || p[0] == '/' || lg.base_dir.empty()) return p;
    return lg.base_dir + "/" + p;
}

static kc_image *pixel_image(float v)
{
    kc_plane *p = plane_new_const(1, 1, v);
    kc_image *img = image_new(1, &p);
    plane_release(p);
    return img;
}

static kc_image *pixel_image_rgba(float r, float g, float b, float a)
{
    kc_plane *p[4] = { plane_new_const(1, 1, r), plane_new_const(1, 1, g), plane_new_const(1, 1, b), plane_new_const(1, 1, a) };
    kc_image *img = image_new(4, p);
    for (auto *q : p) plane_release(q);
    return img;
}

static const SlotData *with_slot(const SlotList &v, uint32_t slot)
{
    for (auto &sd : v)
        if (sd.slot_id == slot) return &sd;
    return nullptr;
}

static int process_graph_node(kc_live_graph &parent, const Node &node, const SlotList &slot_datas, SlotList &out)
{
    // graph::process, src/node/graph.rs:14-51
    if (parent.depth > 32) {
        set_error("Graph nodes nested too deeply");
        return KC_ERR_NODE_PROCESSING;
    }
    kc_live_graph child;
    child.tp = parent.tp;
    child.depth = parent.depth + 1;
    child.base_dir = parent.base_dir;
    if (node.graph) child.g = *node.graph;
    child.reset_node_states();
    for (auto &sd : slot_datas) {
        image_retain(sd.image);
        child.input_slot_datas.push_back(SlotData{ sd.slot_id, 0, sd.image });  // NodeId(slot_id), SlotId(0)
    }
    for (uint32_t oid : child.g.output_ids()) {
        KC_TRY(child.await_clean(oid));
        for (auto &sd : child.slots_of(oid)) {
            image_retain(sd.image);
            out.push_back(SlotData{ node.node_id, oid, sd.image });
        }
    }
    return KC_OK;
}

static int dispatch(kc_live_graph &lg, const Node &node, const SlotList &sd, SlotList &out)
{
    // process_node_internal, src/node/node_type.rs:98-138
    const uint32_t nid = node.node_id;
    switch (node.type) {
    case KC_NODE_INPUT_RGBA: {
        // input_rgba::process, src/node/input_rgba.rs:7-13: takes input_node_datas[0]
        if (lg.input_slot_datas.empty()) {
            set_error("InputRgba without input slot data (index out of bounds in the reference)");
            return KC_ERR_NODE_PROCESSING;
        }
        image_retain(lg.input_slot_datas[0].image);
        out.push_back(SlotData{ nid, 0, lg.input_slot_datas[0].image });
        return KC_OK;
    }
    case KC_NODE_INPUT_GRAY:
        // input_gray::process, src/node/input_gray.rs:7-16
        for (auto &in : lg.input_slot_datas)
            if (in.node_id == nid) {
                image_retain(in.image);
                out.push_back(SlotData{ in.node_id, in.slot_id, in.image });
                break;
            }
        return KC_OK;
    case KC_NODE_OUTPUT_GRAY:
    case KC_NODE_OUTPUT_RGBA:
        // output::process, src/node/output.rs:12-33
        if (!sd.empty()) {
            image_retain(sd[0].image);
            out.push_back(SlotData{ nid, 0, sd[0].image });
        } else if (node.type == KC_NODE_OUTPUT_RGBA) {
            out.push_back(SlotData{ nid, 0, pixel_image_rgba(0.0f, 0.0f, 0.0f, 1.0f) });
        } else {
            out.push_back(SlotData{ nid, 0, pixel_image(0.0f) });
        }
        return KC_OK;
    case KC_NODE_GRAPH: return process_graph_node(lg, node, sd, out);
    case KC_NODE_IMAGE: {
        // image::process, src/node/image.rs:10-26: unreadable file -> 1x1 magenta
        kc_image *img = nullptr;
        std::vector<uint8_t> px;
        uint32_t w = 0, h = 0;
        int ch = 0;
        if (png_read(resolve_path(lg, node.text), px, w, h, ch) == KC_OK) {
            KC_TRY(image_from_u8(px.data(), w, h, ch, &img));
        } else {
            img = pixel_image_rgba(1.0f, 0.0f, 1.0f, 1.0f);
        }
        out.push_back(SlotData{ nid, 0, img });
        return KC_OK;
    }
    case KC_NODE_EMBED:
        // embed::process, src/node/embed.rs:33-50
        for (auto &e : lg.embedded)
            if (e.slot_data_id == node.embed_id) {
                image_retain(e.image);
                out.push_back(SlotData{ nid, 0, e.image });
                return KC_OK;
            }
        set_error("embedded slot data not found");
        return KC_ERR_NODE_PROCESSING;
    case KC_NODE_WRITE: {
        // write::process, src/node/write.rs:5-21
        if (!sd.empty()) {
            kc_image *img = sd[0].image;
            std::vector<uint8_t> px((size_t)img->w() * img->h() * 4);
            KC_TRY(image_to_u8(img, false, px.data()));
            KC_TRY(png_write_rgba8(resolve_path(lg, node.text), px.data(), img->w(), img->h()));
        }
        return KC_OK;
    }
    case KC_NODE_VALUE: {
        kc_image *img = nullptr;
        KC_TRY(value_process(node.value, &img));
        out.push_back(SlotData{ nid, 0, img });
        return KC_OK;
    }
    case KC_NODE_MIX: {
        const SlotData *l = with_slot(sd, 0), *r = with_slot(sd, 1);
        kc_image *img = nullptr;
        KC_TRY(mix_process(l ? l->image : nullptr, r ? r->image : nullptr, node.mix_type, &img));
        if (img) out.push_back(SlotData{ nid, 0, img });
        return KC_OK;
    }
    case KC_NODE_HEIGHT_TO_NORMAL: {
        const SlotData *in = with_slot(sd, 0);
        kc_image *img = nullptr;
        KC_TRY(height_to_normal_process(in ? in->image : nullptr, &img));
        if (img) out.push_back(SlotData{ nid, 0, img });
        return KC_OK;
    }
    case KC_NODE_SEPARATE_RGBA: {
        kc_image *o[4];
        KC_TRY(separate_process(sd.empty() ? nullptr : sd[0].image, o));  // slot_datas.get(0)
        for (uint32_t i = 0; i < 4; ++i) out.push_back(SlotData{ nid, i, o[i] });
        return KC_OK;
    }
    case KC_NODE_COMBINE_RGBA: {
        kc_image *in[4];
        for (uint32_t i = 0; i < 4; ++i) {
            const SlotData *s = with_slot(sd, i);
            in[i] = s ? s->image : nullptr;
        }
        kc_image *img = nullptr;
        KC_TRY(combine_process(in, &img));
        out.push_back(SlotData{ nid, 0, img });
        return KC_OK;
    }
    }
    set_error("invalid NodeType");
    return KC_ERR_INVALID_NODE_TYPE;
}

int process_node(kc_live_graph &lg, const Node &node, const SlotList &inputs, const std::vector<kc_edge> &edges,
                 SlotList &out)
{
    if (edges.size() != inputs.size()) {
        set_error("process_node: edges / slot data count mismatch");
        return KC_ERR_INVALID_BUFFER_COUNT;
    }
    // node_type.rs:229-231: edges sorted by input slot (slot datas stay in edge insertion order)
    // (a stable insertion sort: a node has a handful of edges, and std::stable_sort takes a heap buffer)
    EdgeList sorted(edges.data(), edges.size());
    for (size_t i = 1; i < sorted.size(); ++i) {
        const kc_edge e = sorted[i];
        size_t j = i;
        for (; j > 0 && sorted[j - 1].input_slot > e.input_slot; --j) sorted[j] = sorted[j - 1];
        sorted[j] = e;
    }

    // resize_buffers, src/shared.rs:141-216
    SlotList resized;
    KC_PROF("process_node_body");
    if (!inputs.empty()) {
        SmallVec<kc_size, 8> sizes;
        for (auto &sd : inputs) sizes.push_back(kc_size{ sd.image->w(), sd.image->h() });
        int slot_index = -1;
        if (node.policy == KC_POLICY_SPECIFIC_SLOT) {
            // shared.rs:113-131
            const kc_edge *edge = nullptr;
            for (auto &e : sorted)
                if (e.input_slot == node.policy_slot) {
                    edge = &e;
                    break;
                }
            if (!edge && !sorted.empty()) edge = &sorted[0];
            if (edge)
                for (size_t i = 0; i < inputs.size(); ++i)
                    if (inputs[i].slot_id == edge->output_slot && inputs[i].node_id == edge->output_id) {
                        slot_index = (int)i;
                        break;
                    }
        }
        kc_size size;
        KC_TRY(calculate_size(node.policy, sizes.data(), (int)sizes.size(), slot_index, node.policy_size, &size));
        for (auto &sd : inputs) {
            kc_image *img = nullptr;
            if (sd.image->w() != size.width || sd.image->h() != size.height) {
                int s = resize_image(sd.image, size, node.filter, &img);
                if (s != KC_OK) {
                    release_all(resized);
                    return s;
                }
            } else {
                img = sd.image;
                image_retain(img);
            }
            resized.push_back(SlotData{ sd.node_id, sd.slot_id, img });
        }
    }
    // assign_slot_ids, node_type.rs:250-267
    SlotList assigned;
    for (auto &e : sorted)
        for (auto &sd : resized)
            if (e.output_slot == sd.slot_id && e.output_id == sd.node_id) {
                image_retain(sd.image);
                assigned.push_back(SlotData{ e.input_id, e.input_slot, sd.image });
                break;
            }
    release_all(resized);

    SlotList &result = out;  // empty on entry; stays empty on every error return
    result.clear();
    int s;
    {
        KC_PROF("dispatch");
        s = dispatch(lg, node, assigned, result);
    }
    release_all(assigned);
    if (s != KC_OK) {
        release_all(result);
        return s;
    }
    if (!node.is_output() && result.size() != node_output_slot_count(node)) {
        // node_type.rs:124-137
        set_error("the number of output buffers does not match the number of output slots");
        release_all(result);
        return KC_ERR_INVALID_BUFFER_COUNT;
    }
    return KC_OK;
}

}  // namespace kc

// ------------------------------------------------------------------------------------------
// LiveGraph, src/live_graph.rs
// ------------------------------------------------------------------------------------------
using namespace kc;

kc_live_graph::~kc_live_graph()
{
    replay_clear();
    clear_data();
    for (auto &e : embedded) image_release(e.image);
    for (auto &i : input_slot_datas) image_release(i.image);
}

void kc_live_graph::clear_data()
{
    for (auto &kv : slot_datas)
        for (auto &sd : kv.second) image_release(sd.image);
    slot_datas.clear();
}

void kc_live_graph::remove_nodes_data(uint32_t id)
{
    // :539-548
    auto it = slot_datas.find(id);
    if (it == slot_datas.end()) return;
    for (auto &sd : it->second) image_release(sd.image);
    slot_datas.erase(it);
}

const kc::SmallVec<kc::SlotData, 4> &kc_live_graph::slots_of(uint32_t node) const
{
    static const kc::SmallVec<kc::SlotData, 4> none{};
    auto it = slot_datas.find(node);
    return it == slot_datas.end() ? none : it->second;
}

const SlotData *kc_live_graph::find_slot(uint32_t node, uint32_t slot) const
{
    for (auto &sd : slots_of(node))
        if (sd.slot_id == slot) return &sd;
    return nullptr;
}

int kc_live_graph::state_of(uint32_t id, int *st) const
{
    auto it = node_state.find(id);
    if (it == node_state.end()) return KC_ERR_INVALID_NODE_ID;
    *st = it->second;
    return KC_OK;
}

int kc_live_graph::set_state(uint32_t id, int st)
{
    // :515-537.  Dirty propagates to the children; a node that already is Dirty ends the walk.  The reference
    // recurses before it stores the node's own state, which never terminates on a cyclic graph: here the
    // state is stored first and the children go on a worklist (same final states on every acyclic graph).
    int old;
    KC_TRY(state_of(id, &old));
    if (st == old) return KC_OK;
    if (st != KC_STATE_DIRTY) {
        node_state[id] = st;
        changed.insert(id);
        return KC_OK;
    }
    SmallVec<uint32_t, 64> work;
    work.push_back(id);
    while (!work.empty()) {
        const uint32_t n = work[work.size() - 1];
        work.erase_at(work.size() - 1);
        auto st = node_state.find(n);
        if (st == node_state.end()) return KC_ERR_INVALID_NODE_ID;
        const int cur = st->second;
        if (cur == KC_STATE_DIRTY) continue;
        st->second = cur == KC_STATE_PROCESSING ? KC_STATE_PROCESSING_DIRTY : KC_STATE_DIRTY;
        changed.insert(n);
        // straight from the edge index: a child listed twice (two edges from n) is skipped by the state test above
        // (get_children would allocate, sort and de-duplicate a vector per node: a third of a connect() on a 32-node chain)
        for (auto &e : g.edges_out_of(n)) work.push_back(e.input_id);
    }
    return KC_OK;
}

int kc_live_graph::force_state(uint32_t id, int st)
{
    KC_TRY(set_state(id, st));
    node_state[id] = st;
    return KC_OK;
}

void kc_live_graph::reset_node_states()
{
    node_state.clear();
    for (auto &n : g.nodes) node_state[n.node_id] = KC_STATE_DIRTY;
}

int kc_live_graph::add_node(Node n, uint32_t *id)
{
    uint32_t nid = 0;
    KC_TRY(g.add_node(std::move(n), &nid));
    changed.insert(nid);
    node_state[nid] = KC_STATE_DIRTY;
    if (id) *id = nid;
    return KC_OK;
}

int kc_live_graph::add_node_with_id(Node n)
{
    uint32_t nid = n.node_id;
    KC_TRY(g.add_node_with_id(std::move(n)));
    changed.insert(nid);
    node_state[nid] = KC_STATE_DIRTY;
    return KC_OK;
}

int kc_live_graph::remove_node(uint32_t id)
{
    // :452-475
    std::vector<kc_edge> removed;
    KC_TRY(g.remove_node(id, &removed));
    changed.insert(id);
    for (auto &e : removed) changed.insert(e.input_id);
    remove_nodes_data(id);
    node_state.erase(id);
    return KC_OK;
}

int kc_live_graph::connect(uint32_t on, uint32_t in, uint32_t os, uint32_t is)
{
    KC_PROF("lg_connect");
    // :488-511
    KC_TRY(g.connect(on, in, os, is));
    changed.insert(in);
    KC_TRY(set_state(in, KC_STATE_DIRTY));
    return KC_OK;
}

int kc_live_graph::remove_edge(kc_edge e)
{
    // :551-566
    if (!g.find(e.input_id)) return KC_ERR_INVALID_NODE_ID;
    std::vector<uint32_t> dirty = g.get_children_recursive(e.input_id);
    dirty.push_back(e.input_id);
    std::sort(dirty.begin(), dirty.end());
    dirty.erase(std::unique(dirty.begin(), dirty.end()), dirty.end());
    KC_TRY(g.remove_edge(e));
    for (uint32_t id : dirty) {
        KC_TRY(set_state(id, KC_STATE_DIRTY));
        remove_nodes_data(id);
    }
    return KC_OK;
}

int kc_live_graph::disconnect_slot(uint32_t id, int side, uint32_t slot)
{
    // :568-594
    std::vector<kc_edge> removed;
    KC_TRY(g.disconnect_slot(id, side, slot, &removed));
    std::vector<uint32_t> dirty;
    for (auto &e : removed) {
        auto sub = g.get_children_recursive(e.input_id);
        dirty.insert(dirty.end(), sub.begin(), sub.end());
    }
    if (side == KC_SIDE_INPUT)
        dirty.push_back(id);
    else
        changed.insert(id);
    // the reference (:576-583) marks children_recursive(edge.input_id) -- which excludes the
    // consumer itself on the Output side; the consumer keeps its state there too
    std::sort(dirty.begin(), dirty.end());
    dirty.erase(std::unique(dirty.begin(), dirty.end()), dirty.end());
    for (uint32_t d : dirty) KC_TRY(set_state(d, KC_STATE_DIRTY));
    return KC_OK;
}

// One node through process_node, then the bookkeeping of src/engine.rs:34-103.
int kc_live_graph::process_one(uint32_t id)
{
    KC_PROF("process_one");
    const Node *np = g.find(id);
    if (!np) return KC_ERR_INVALID_NODE_ID;
    // Nothing below changes nodes or edges (operators only read them, a Graph node evaluates a copy of its
    // child graph), so the node and its edge list are used in place.
    const Node &node = *np;
    node_state[id] = KC_STATE_PROCESSING;
    // engine.rs:213-218: input edges in insertion order; :261-275: one SlotData per edge
    const std::vector<kc_edge> &edges = g.edges_into(id);
    SlotList inputs;
    for (auto &e : edges) {
        const SlotData *sd = find_slot(e.output_id, e.output_slot);
        if (!sd) {
            set_error("input slot data missing");
            node_state[id] = KC_STATE_DIRTY;
            return KC_ERR_NO_SLOT_DATA;
        }
        inputs.push_back(*sd);
    }
    SlotList outs;
    int s = process_node(*this, node, inputs, edges, outs);
    if (s != KC_OK) {
        // engine.rs:111-118 shuts the processor down and panics; here the node goes back to Dirty
        node_state[id] = KC_STATE_DIRTY;
        return s;
    }
    // use_cache keeps every node's planes, so they are materialised now (one kernel per node);
    // otherwise they stay lazy and are dropped or fused by whoever consumes them.
    if (use_cache)
        for (auto &sd : outs) {
            int fs = image_force(sd.image);
            if (fs != KC_OK) {
                for (auto &o : outs) image_release(o.image);
                node_state[id] = KC_STATE_DIRTY;
                return fs;
            }
        }
    remove_nodes_data(id);
    for (auto &sd : outs) slot_datas[sd.node_id].push_back(sd);
    if (!use_cache) {
        // engine.rs:58-75: a parent's planes are dropped once every child of it is Clean or Processing
        // (edge lists straight from the index: a parent or child seen twice changes nothing)
        for (auto &pe : edges) {
            // the scan runs from the newest edge: children are usually evaluated in the order they were connected,
            // so a child that is still Dirty is found at once
            bool all_done = true;
            const std::vector<kc_edge> &children = g.edges_out_of(pe.output_id);
            for (size_t i = children.size(); all_done && i-- > 0;) {
                int st = KC_STATE_DIRTY;
                (void)state_of(children[i].input_id, &st);
                all_done = st == KC_STATE_CLEAN || st == KC_STATE_PROCESSING;
            }
            if (all_done) remove_nodes_data(pe.output_id);
        }
    }
    return set_state(id, KC_STATE_CLEAN);
}

int kc_live_graph::ensure_clean(uint32_t root)
{
    KC_PROF("ensure_clean_total");
    // Parents first (LiveGraph::get_closest_processable, :279-311, collapsed into a depth-first walk).  The
    // walk keeps its own stack -- a 100 000-node chain must not exhaust the thread's.  connect() and the JSON
    // reader accept a cycle (as the reference's do) and a node on one can never become Clean: the stack is a
    // path of distinct nodes otherwise, so one deeper than the node count has walked round a cycle.
    struct Frame {
        uint32_t id;
        const std::vector<kc_edge> *edges;  // the index's list: nothing changes nodes or edges during the walk
        size_t next;
        bool awaiting;  // edges[next - 1]'s producer has just been brought up to date
    };
    std::vector<Frame> stack;
    stack.reserve(64);
    const size_t max_depth = g.nodes.size();
    auto enter = [&](uint32_t id) -> int {
        int st;
        KC_TRY(state_of(id, &st));
        if (st == KC_STATE_CLEAN) return KC_OK;
        if (stack.size() >= max_depth) {
            std::set<uint32_t> seen;
            uint32_t twice = id;
            for (auto &f : stack)
                if (!seen.insert(f.id).second) {
                    twice = f.id;
                    break;
                }
            set_error("graph has a cycle through node " + std::to_string(twice));
            return KC_ERR_NODE_PROCESSING;
        }
        stack.push_back(Frame{ id, &g.edges_into(id), 0, false });
        return KC_OK;
    };
    KC_TRY(enter(root));
    while (!stack.empty()) {
        Frame &f = stack.back();
        if (f.awaiting) {
            const kc_edge &e = (*f.edges)[f.next - 1];
            f.awaiting = false;
            if (!find_slot(e.output_id, e.output_slot)) {
                set_error("a parent produced no data for a connected slot");
                return KC_ERR_NO_SLOT_DATA;
            }
        }
        if (f.next == f.edges->size()) {
            const uint32_t id = f.id;
            stack.pop_back();
            KC_TRY(process_one(id));
            continue;
        }
        const kc_edge e = (*f.edges)[f.next++];
        int pst;
        if (state_of(e.output_id, &pst) != KC_OK) continue;  // parent deleted
        if (pst == KC_STATE_CLEAN && !find_slot(e.output_id, e.output_slot)) {
            // engine.rs:264-271: data was dropped (use_cache == false) -> parent goes Dirty again
            const Node *pn = g.find(e.output_id);
            bool has_slot = false;
            if (pn)
                for (auto &sl : node_output_slots(*pn)) has_slot |= sl.slot_id == e.output_slot;
            if (!has_slot) return KC_ERR_NO_SLOT_DATA;
            // Only the parent itself: it is recomputed right below and, planes being a pure function of the graph, yields
            // the data it had.  The reference's set_state would dirty every descendant too and its engine then recomputes
            // them; this walk has already passed some of them (another parent of the node on the stack, brought up to
            // date a moment ago) and would have left them Dirty WITH Clean children -- after which an edit of such a node
            // (set_mix_type: "already Dirty", no propagation) did not reach its children (profiles/soak_fuzz.py, edit
            // seeds 210 and 5678).
            node_state[e.output_id] = KC_STATE_DIRTY;
            changed.insert(e.output_id);
        }
        f.awaiting = true;
        KC_TRY(enter(e.output_id));  // may invalidate f
    }
    return KC_OK;
}

int kc_live_graph::await_clean(uint32_t id)
{
    KC_PROF("await_clean_total");
    if (!g.find(id)) return KC_ERR_INVALID_NODE_ID;
    // an evaluation that repeats the recorded one exactly is replayed without the walk (replay.cpp)
    bool replayed = false;
    KC_TRY(replay_try(*this, id, &replayed));
    if (replayed) return KC_OK;
    // A graph's FIRST evaluation builds its chains the way every kernel can run them (4 input planes, no joined chains) when
    // there is no chance of finding a compiled kernel for anything else: a process that evaluates a graph once must not pay for
    // programs whose own kernels it will never see compiled.  From the second evaluation on the chains are built for those
    // kernels (csrc/runtime.cpp chain_in_limit / join_ok); "compile at first sight" (kc_set_specialize(2)) has them from the
    // start, and so has a process that finds code objects of earlier processes or of the build on disk (specialize.cpp,
    // kernel cache) -- where a program's kernel is missing all the same, chain_launch cuts the chain as before.
    Context &c = ctx();
    const bool plain_before = c.plain_chains;
    c.plain_chains = plain_before || (walks == 0 && specialize_get_mode() == 1 && !kernel_cache_populated());
    ReplayRecorder *rec = replay_begin(*this, id);
    const int s = await_clean_walk(id);
    c.plain_chains = plain_before;
    ++walks;
    replay_end(*this, id, rec, s);
    return s;
}

int kc_live_graph::await_clean_walk(uint32_t id)
{
    ResizeMemoScope memo;
    if (auto_update) KC_TRY(update());
    KC_TRY(ensure_clean(id));
    // Clean means computed: whatever the node still holds is brought into HBM now.
    for (auto &sd : slots_of(id)) KC_TRY(image_force(sd.image));
    return KC_OK;
}

int kc_live_graph::update()
{
    ResizeMemoScope memo;
    // engine.rs:131-167
    std::vector<uint32_t> requested;
    for (auto &kv : node_state) {
        if (auto_update) {
            if (kv.second != KC_STATE_CLEAN && kv.second != KC_STATE_PROCESSING && kv.second != KC_STATE_PROCESSING_DIRTY)
                requested.push_back(kv.first);
        } else if (kv.second == KC_STATE_REQUESTED || kv.second == KC_STATE_PRIORITISED) {
            requested.push_back(kv.first);
        }
    }
    std::sort(requested.begin(), requested.end());  // ascending ids, whatever order the state table iterates in
    for (uint32_t id : requested) {
        if (!g.find(id)) continue;
        KC_TRY(ensure_clean(id));
    }
    for (uint32_t id : requested)
        for (auto &sd : slots_of(id)) KC_TRY(image_force(sd.image));
    return KC_OK;
}
