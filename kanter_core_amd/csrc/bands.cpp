// Row-band evaluation: rows [y0, y1) of a node's result without computing the rest of the image.
//
// This is the data-level way to put several GPUs on ONE graph: every rank evaluates the same
// graph for its own band of the requested node and the bands, stacked, are the full result -- bit for bit, because
// every pixel goes through the same operations in the same order as in the whole-image evaluation:
//   * pointwise nodes (Mix, as_type, Separate / Combine, Output) need the same rows of their inputs;
//   * HeightToNormal reads the pixel above (src/node/height_to_normal.rs:46-52, toroidal: src/node/
//     process_shared.rs:31-65): one more row on top, which for the band starting at row 0 is the image's LAST row;
//   * an implicit resize (src/shared.rs:141-216) needs the source rows its vertical taps read,
//     [left(y0), left(y1 - 1) + count(y1 - 1)), about support * max(in / out, 1) rows beyond the band's image.
// No exchange happens during the evaluation: a band is widened by the halo rows its consumers need and those rows
// are computed redundantly by the neighbouring ranks (a 1-row halo per HeightToNormal, a few rows per resize).
// What the host has to provide is the rows of the SOURCE images (Embed / Image / Input*) the plan asks for --
// kc_live_graph_band_source_rows lists them, so sources that are themselves sharded by rows can be loaded or
// exchanged with exactly their halo (kc_live_graph_embed_slot_data_band).
//
// The walk: (1) ancestors in topological order, (2) logical output size of every node (the same policy code the
// whole-image path runs, on sizes only), (3) needed rows, root to sources, (4) evaluation, sources to root, on
// images that hold only the needed rows.  Rows are logical indices; a negative row r means row r + height (the wrap).
#include <algorithm>

#include "kc_runtime.hpp"

namespace kc {

namespace {

struct Need {
    int32_t a = 0, b = 0;
    bool set = false;
    void add(int32_t lo, int32_t hi)
    {
        if (!set) {
            a = lo;
            b = hi;
            set = true;
        } else {
            a = std::min(a, lo);
            b = std::max(b, hi);
        }
    }
};

// rows [a, b) of a `h`-row image; anything that covers the whole height is the whole image
void normalise(Need &n, uint32_t h)
{
    if (!n.set) return;
    if (h == 1 || (int64_t)n.b - n.a >= (int64_t)h) {
        n.a = 0;
        n.b = (int32_t)h;
    }
}

struct BandSlot {
    uint32_t slot_id;
    kc_image *img;  // retained; holds logical rows [y0, y0 + planes[0]->h)
    int32_t y0;
};

struct Walk {
    kc_live_graph &lg;
    std::vector<uint32_t> topo;
    std::map<uint32_t, kc_size> size;
    std::map<uint32_t, bool> rgba;  // type of the node's output(s): some operators' output SIZE depends on it (Separate of a gray image)
    std::map<uint32_t, Need> need;
    std::map<uint32_t, std::vector<BandSlot>> data;
    explicit Walk(kc_live_graph &g) : lg(g) {}
    ~Walk()
    {
        for (auto &kv : data)
            for (auto &s : kv.second) image_release(s.img);
    }
};

int topo_order(const NodeGraph &g, uint32_t root, std::vector<uint32_t> &topo)
{
    std::map<uint32_t, int> mark;
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
        if (!g.find(p) || mark[p] == 2) continue;
        if (mark[p] == 1) {
            set_error("graph has a cycle through node " + std::to_string(p));
            return KC_ERR_NODE_PROCESSING;
        }
        mark[p] = 1;
        st.push_back(Frame{ p, g.get_parents(p), 0 });
    }
    return KC_OK;
}

bool is_source(const Node &n) { return n.type == KC_NODE_EMBED || n.type == KC_NODE_IMAGE || n.is_input(); }

// Is a source node's image RGBA?  (Image nodes are: image_from_u8 always builds four planes.)
bool source_is_rgba(Walk &W, const Node &n)
{
    kc_live_graph &lg = W.lg;
    if (n.type == KC_NODE_EMBED) {
        for (auto &e : lg.embedded)
            if (e.slot_data_id == n.embed_id) return e.image->is_rgba();
        return false;
    }
    if (n.type == KC_NODE_IMAGE) return true;
    if (n.type == KC_NODE_INPUT_RGBA) return !lg.input_slot_datas.empty() && lg.input_slot_datas[0].image->is_rgba();
    for (auto &in : lg.input_slot_datas)
        if (in.node_id == n.node_id) return in.image->is_rgba();
    return false;
}

// ---- (2) sizes -------------------------------------------------------------------------------------------
int source_size(Walk &W, const Node &n, kc_size *out)
{
    kc_live_graph &lg = W.lg;
    if (n.type == KC_NODE_EMBED) {
        for (auto &e : lg.embedded)
            if (e.slot_data_id == n.embed_id) {
                *out = kc_size{ e.image->w(), e.full_h ? e.full_h : e.image->h() };
                return KC_OK;
            }
        set_error("embedded slot data not found");
        return KC_ERR_NODE_PROCESSING;
    }
    if (n.type == KC_NODE_IMAGE) {
        std::vector<uint8_t> px;
        uint32_t w = 0, h = 0;
        int ch = 0;
        const std::string path = (n.text.empty() || n.text[0] == '/' || lg.base_dir.empty()) ? n.text : lg.base_dir + "/" + n.text;
        *out = png_read(path, px, w, h, ch) == KC_OK ? kc_size{ w, h } : kc_size{ 1, 1 };  // unreadable -> 1x1 magenta
        return KC_OK;
    }
    if (n.type == KC_NODE_INPUT_RGBA) {
        if (lg.input_slot_datas.empty()) {
            set_error("InputRgba without input slot data");
            return KC_ERR_NODE_PROCESSING;
        }
        *out = kc_size{ lg.input_slot_datas[0].image->w(), lg.input_slot_datas[0].image->h() };
        return KC_OK;
    }
    for (auto &in : lg.input_slot_datas)
        if (in.node_id == n.node_id) {
            *out = kc_size{ in.image->w(), in.image->h() };
            return KC_OK;
        }
    set_error("InputGray without input slot data");
    return KC_ERR_NO_SLOT_DATA;
}

// calculate_size of one node from its parents' logical sizes, as process_node does (graph.cpp)
int target_size(Walk &W, const Node &n, const std::vector<kc_edge> &edges, kc_size *out)
{
    std::vector<kc_size> sizes;
    for (auto &e : edges) sizes.push_back(W.size[e.output_id]);
    int slot_index = -1;
    if (n.policy == KC_POLICY_SPECIFIC_SLOT) {
        std::vector<kc_edge> sorted = edges;
        std::stable_sort(sorted.begin(), sorted.end(), [](const kc_edge &x, const kc_edge &y) { return x.input_slot < y.input_slot; });
        const kc_edge *edge = nullptr;
        for (auto &e : sorted)
            if (e.input_slot == n.policy_slot) {
                edge = &e;
                break;
            }
        if (!edge && !sorted.empty()) edge = &sorted[0];
        if (edge)
            for (size_t i = 0; i < edges.size(); ++i)
                if (edges[i].output_slot == edge->output_slot && edges[i].output_id == edge->output_id) {
                    slot_index = (int)i;
                    break;
                }
    }
    return calculate_size(n.policy, sizes.data(), (int)sizes.size(), slot_index, n.policy_size, out);
}

int infer_sizes(Walk &W, const uint32_t *skip = nullptr)
{
    const NodeGraph &g = W.lg.g;
    for (uint32_t id : W.topo) {
        if (skip && id == *skip) continue;
        const Node &n = *g.find(id);
        kc_size s{ 1, 1 };
        bool rgba = false;
        if (is_source(n)) {
            KC_TRY(source_size(W, n, &s));
            rgba = source_is_rgba(W, n);
        } else if (n.type == KC_NODE_VALUE) {
            s = kc_size{ 1, 1 };
        } else if (n.type == KC_NODE_GRAPH || n.type == KC_NODE_WRITE) {
            set_error("row-band evaluation does not go through Write nodes (Graph nodes are expanded before the walk)");
            return KC_ERR_UNSUPPORTED;
        } else {
            const std::vector<kc_edge> &edges = g.edges_into(id);
            if (edges.empty()) {
                if (n.type == KC_NODE_HEIGHT_TO_NORMAL) {
                    set_error("HeightToNormal without input produces no data");
                    return KC_ERR_NO_SLOT_DATA;
                }
                s = kc_size{ 1, 1 };
            } else {
                KC_TRY(target_size(W, n, edges, &s));
            }
            // the operators' own rules (csrc/ops.cpp), as far as they decide the size or the type of what comes out
            auto parent_on = [&](uint32_t slot) -> const kc_edge * {
                for (auto &e : edges)
                    if (e.input_slot == slot) return &e;
                return nullptr;
            };
            switch (n.type) {
            case KC_NODE_MIX: {
                const kc_edge *l = parent_on(0), *r = parent_on(1);
                rgba = l ? W.rgba[l->output_id] : r ? W.rgba[r->output_id] : false;  // as_type(right, left's type)
                break;
            }
            case KC_NODE_SEPARATE_RGBA: {
                // slot_datas.get(0): the input on the lowest connected slot; a gray (or missing) one gives four 1x1 zeros
                const kc_edge *first = nullptr;
                for (auto &e : edges)
                    if (!first || e.input_slot < first->input_slot) first = &e;
                if (!first || !W.rgba[first->output_id]) s = kc_size{ 1, 1 };
                rgba = false;
                break;
            }
            case KC_NODE_COMBINE_RGBA:
            case KC_NODE_HEIGHT_TO_NORMAL: rgba = true; break;
            case KC_NODE_OUTPUT_GRAY:
            case KC_NODE_OUTPUT_RGBA: {
                const kc_edge *first = nullptr;
                for (auto &e : edges)
                    if (!first || e.input_slot < first->input_slot) first = &e;
                rgba = first ? W.rgba[first->output_id] : n.type == KC_NODE_OUTPUT_RGBA;
                break;
            }
            default: break;
            }
        }
        W.size[id] = s;
        W.rgba[id] = rgba;
    }
    return KC_OK;
}

// ---- (3) needs -------------------------------------------------------------------------------------------
// source rows the vertical taps of output rows [a, b) read (a >= 0: no wrap)
int resize_source_rows(uint32_t in_h, uint32_t out_h, int filter, int32_t a, int32_t b, int32_t *sa, int32_t *sb)
{
    // The planner asks this once per resize edge per call, and the stateless band path calls the planner every step: the
    // windows (host only: no device needed here) are kept per (in, out, filter) instead of rebuilt -- O(image height) each.
    // Callers hold the context lock.
    static std::map<std::tuple<uint32_t, uint32_t, int>, std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> windows;
    auto key = std::make_tuple(in_h, out_h, filter);
    auto it = windows.find(key);
    if (it == windows.end()) {
        TapsHost th;
        KC_TRY(build_taps_host(in_h, out_h, filter, th));
        if (windows.size() >= 256) windows.clear();
        it = windows.emplace(key, std::make_pair(std::move(th.left), std::move(th.count))).first;
    }
    struct {
        const std::vector<uint32_t> &left, &count;
    } t{ it->second.first, it->second.second };
    int64_t lo = in_h, hi = 0;
    for (int32_t y = a; y < b; ++y) {
        lo = std::min<int64_t>(lo, t.left[(size_t)y]);
        hi = std::max<int64_t>(hi, (int64_t)t.left[(size_t)y] + t.count[(size_t)y]);
    }
    *sa = (int32_t)lo;
    *sb = (int32_t)hi;
    return KC_OK;
}

int propagate_needs(Walk &W, uint32_t root, int32_t y0, int32_t y1)
{
    const NodeGraph &g = W.lg.g;
    const kc_size rs = W.size[root];
    if (rs.height == 1) {
        y0 = 0;
        y1 = 1;
    }
    if (y0 < 0 || y1 <= y0 || (uint32_t)y1 > rs.height) {
        set_error("row band outside the node's image (height " + std::to_string(rs.height) + ")");
        return KC_ERR_INVALID_ARG;
    }
    W.need[root].add(y0, y1);
    for (size_t k = W.topo.size(); k-- > 0;) {
        const uint32_t id = W.topo[k];
        Need &nd = W.need[id];
        if (!nd.set) continue;
        const Node &n = *g.find(id);
        const kc_size T = W.size[id];
        normalise(nd, T.height);
        if (is_source(n) || n.type == KC_NODE_VALUE) continue;
        Need in = nd;
        if (n.type == KC_NODE_HEIGHT_TO_NORMAL && T.height > 1) in.a -= 1;  // the row above; row -1 wraps to the last row
        if (n.type == KC_NODE_HEIGHT_TO_NORMAL && T.height == 1) in = Need{ 0, 1, true };
        for (auto &e : g.edges_into(id)) {
            const kc_size ps = W.size[e.output_id];
            Need &pn = W.need[e.output_id];
            if (ps.width == T.width && ps.height == T.height) {
                pn.add(in.a, in.b);
            } else if (ps.height == 1 || in.a < 0 || (uint32_t)in.b > T.height) {
                pn.add(0, (int32_t)ps.height);  // wrapped or degenerate: the whole (small) source
            } else {
                int32_t sa = 0, sb = 0;
                KC_TRY(resize_source_rows(ps.height, T.height, n.filter, in.a, in.b, &sa, &sb));
                pn.add(sa, sb);
            }
        }
    }
    return KC_OK;
}

// ---- (4) evaluation --------------------------------------------------------------------------------------
// a view of rows [off, off + rows) of a resident plane: no bytes move, the parent stays alive with the view
kc_plane *plane_rows_view(kc_plane *p, uint32_t off, uint32_t rows)
{
    kc_plane *v = new kc_plane();
    v->w = p->w;
    v->h = rows;
    v->kind = kc_plane::MEM;
    v->dptr = (float *)((char *)p->dptr + (size_t)off * p->pitch);
    v->pitch = p->pitch;
    v->owned = false;
    v->view_of = p;
    plane_retain(p);
    return v;
}

// rows [a, b) (logical; negative = wrapped) of an image that holds logical rows [y0, y0 + rows) of a `full_h`-row image
int crop_rows(kc_image *img, int32_t y0, uint32_t full_h, int32_t a, int32_t b, kc_image **out)
{
    const uint32_t have = img->planes[0]->h, rows = (uint32_t)(b - a);
    if (full_h == 1) {  // 1x1 images (Values, defaults) are what they are
        image_retain(img);
        *out = img;
        return KC_OK;
    }
    const int64_t off = (int64_t)a - y0;
    const bool inside = off >= 0 && off + rows <= have;
    const bool wrap = !inside && y0 == 0 && have == full_h && a < 0 && b >= 0 && (uint32_t)(-a) <= full_h && (uint32_t)b <= full_h;
    if (!inside && !wrap) {
        set_error("row band: an input does not hold the rows its consumer needs (rows " + std::to_string(a) + ".." + std::to_string(b) +
                  " of an image holding " + std::to_string(y0) + ".." + std::to_string((int64_t)y0 + have) + ")");
        return KC_ERR_INVALID_ARG;
    }
    if (inside && off == 0 && rows == have) {
        image_retain(img);
        *out = img;
        return KC_OK;
    }
    Context &c = ctx();
    kc_plane *p[4] = { nullptr, nullptr, nullptr, nullptr };
    int s = KC_OK;
    for (int i = 0; i < img->n && s == KC_OK; ++i) {
        kc_plane *src = img->planes[i];
        for (int j = 0; j < i; ++j)
            if (img->planes[j] == src) {
                p[i] = p[j];
                plane_retain(p[i]);
                break;
            }
        if (p[i]) continue;
        if (src->kind == kc_plane::CONST) {
            p[i] = plane_new_const(src->w, rows, src->cval);
            continue;
        }
        s = plane_force(src);  // LAZY / RESIZE -> MEM
        if (s != KC_OK) break;
        if (inside) {
            p[i] = plane_rows_view(src, (uint32_t)off, rows);
        } else {
            // toroidal: rows full_h + a .. full_h - 1, then 0 .. b - 1, gathered into a fresh plane
            s = plane_new_mem(src->w, rows, &p[i]);
            if (s != KC_OK) break;
            const uint32_t top = (uint32_t)(-a);
            const size_t wbytes = (size_t)src->w * 4;
            hipError_t e = hipMemcpy2DAsync(p[i]->dptr, p[i]->pitch, (char *)src->dptr + (size_t)(full_h - top) * src->pitch, src->pitch,
                                            wbytes, top, hipMemcpyDeviceToDevice, c.stream);
            if (e == hipSuccess && b > 0)
                e = hipMemcpy2DAsync((char *)p[i]->dptr + (size_t)top * p[i]->pitch, p[i]->pitch, src->dptr, src->pitch, wbytes,
                                     (size_t)b, hipMemcpyDeviceToDevice, c.stream);
            if (e != hipSuccess) s = hip_fail(e, "row band gather");
        }
    }
    if (s == KC_OK) *out = image_new(img->n, p);
    for (int i = 0; i < img->n; ++i) plane_release(p[i]);
    return s;
}

// rows [a, b) of `img` (logical rows [y0, ...) of a src.height-row image) resampled to T: the band form of resize_image
int resize_band(kc_image *img, int32_t y0, kc_size src, kc_size T, int filter, int32_t a, int32_t b, kc_image **out)
{
    const uint32_t rows = (uint32_t)(b - a);
    kc_plane *p[4] = { nullptr, nullptr, nullptr, nullptr };
    kc_plane *srcs[4], *outs[4];
    int idx[4], n = 0, s = KC_OK;
    for (int i = 0; i < img->n && s == KC_OK; ++i) {
        kc_plane *pl = img->planes[i];
        bool alias = false;
        for (int j = 0; j < i; ++j)
            if (img->planes[j] == pl) {
                idx[i] = idx[j];
                alias = true;
                if (idx[j] < 0) {
                    p[i] = p[j];
                    plane_retain(p[i]);
                }
                break;
            }
        if (alias) continue;
        if (src.width == 1 && src.height == 1 && pl->kind == kc_plane::CONST) {
            // a 1x1 source is one tap of weight 1 in both passes, then the clamp (resize.cpp, resize_plane_uncached)
            float t = 0.0f;
            t += pl->cval * 1.0f;
            float u = 0.0f;
            u += t * 1.0f;
            u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
            p[i] = plane_new_const(T.width, rows, u);
            idx[i] = -1;
            continue;
        }
        s = plane_materialize(pl);  // constants of real size are resampled like any plane (as the whole-image path does)
        if (s != KC_OK) break;
        idx[i] = n;
        srcs[n++] = pl;
    }
    if (s == KC_OK && n > 0) s = resize_planes_band(srcs, n, y0, src.height, T, a, b, filter, outs);
    if (s == KC_OK) {
        for (int i = 0; i < img->n; ++i)
            if (!p[i]) {
                p[i] = outs[idx[i]];
                plane_retain(p[i]);
            }
        for (int k = 0; k < n; ++k) plane_release(outs[k]);
        *out = image_new(img->n, p);
    }
    for (int i = 0; i < img->n; ++i) plane_release(p[i]);
    return s;
}

const BandSlot *find_band_slot(const Walk &W, uint32_t node, uint32_t slot)
{
    auto it = W.data.find(node);
    if (it == W.data.end()) return nullptr;
    for (auto &s : it->second)
        if (s.slot_id == slot) return &s;
    return nullptr;
}

int evaluate_source(Walk &W, const Node &n)
{
    kc_live_graph &lg = W.lg;
    const Need nd = W.need[n.node_id];
    const kc_size sz = W.size[n.node_id];
    if (n.type == KC_NODE_EMBED) {
        for (auto &e : lg.embedded)
            if (e.slot_data_id == n.embed_id) {
                kc_image *img = nullptr;
                KC_TRY(crop_rows(e.image, e.full_h ? e.band_y0 : 0, sz.height, nd.a, nd.b, &img));
                W.data[n.node_id].push_back(BandSlot{ 0, img, sz.height == 1 ? 0 : nd.a });
                return KC_OK;
            }
        set_error("embedded slot data not found");
        return KC_ERR_NODE_PROCESSING;
    }
    // Image / Input*: through the ordinary evaluator (whole image), then the needed rows
    KC_TRY(lg.ensure_clean(n.node_id));
    for (auto &sd : lg.slots_of(n.node_id)) {
        kc_image *img = nullptr;
        KC_TRY(crop_rows(sd.image, 0, sz.height, nd.a, nd.b, &img));
        W.data[n.node_id].push_back(BandSlot{ sd.slot_id, img, sz.height == 1 ? 0 : nd.a });
    }
    return KC_OK;
}

int evaluate_node(Walk &W, const Node &n)
{
    const NodeGraph &g = W.lg.g;
    const uint32_t id = n.node_id;
    const kc_size T = W.size[id];
    const Need nd = W.need[id];
    Need in = nd;
    if (n.type == KC_NODE_HEIGHT_TO_NORMAL) in = T.height > 1 ? Need{ nd.a - 1, nd.b, true } : Need{ 0, 1, true };
    std::vector<kc_edge> edges = g.edges_into(id);
    std::vector<kc_edge> sorted = edges;
    std::stable_sort(sorted.begin(), sorted.end(), [](const kc_edge &x, const kc_edge &y) { return x.input_slot < y.input_slot; });
    // resize_buffers on bands, then assign_slot_ids (node_type.rs:229-267): one image per sorted edge
    std::vector<SlotData> assigned;
    int s = KC_OK;
    for (auto &e : sorted) {
        const BandSlot *bs = find_band_slot(W, e.output_id, e.output_slot);
        if (!bs) {
            set_error("a parent produced no data for a connected slot");
            s = KC_ERR_NO_SLOT_DATA;
            break;
        }
        const kc_size ps = W.size[e.output_id];
        kc_image *img = nullptr;
        if (ps.width == T.width && ps.height == T.height)
            s = crop_rows(bs->img, bs->y0, T.height, in.a, in.b, &img);
        else
            s = resize_band(bs->img, bs->y0, ps, T, n.filter, T.height == 1 ? 0 : in.a, T.height == 1 ? 1 : in.b, &img);
        if (s != KC_OK) break;
        assigned.push_back(SlotData{ e.input_id, e.input_slot, img });
    }
    auto with = [&](uint32_t slot) -> kc_image * {
        for (auto &sd : assigned)
            if (sd.slot_id == slot) return sd.image;
        return nullptr;
    };
    std::vector<BandSlot> &out = W.data[id];
    const int32_t oy = T.height == 1 ? 0 : nd.a;
    if (s == KC_OK) switch (n.type) {
        case KC_NODE_MIX: {
            kc_image *img = nullptr;
            s = mix_process(with(0), with(1), n.mix_type, &img);
            if (s == KC_OK && img) out.push_back(BandSlot{ 0, img, oy });
            break;
        }
        case KC_NODE_HEIGHT_TO_NORMAL: {
            kc_image *img = nullptr;
            kc_image *src = with(0);
            if (T.height > 1) s = height_to_normal_band(src, T.height, &img);
            else s = height_to_normal_process(src, &img);
            if (s == KC_OK && img) out.push_back(BandSlot{ 0, img, oy });
            break;
        }
        case KC_NODE_SEPARATE_RGBA: {
            kc_image *o[4];
            s = separate_process(assigned.empty() ? nullptr : assigned[0].image, o);
            if (s == KC_OK)
                for (uint32_t i = 0; i < 4; ++i) out.push_back(BandSlot{ i, o[i], oy });
            break;
        }
        case KC_NODE_COMBINE_RGBA: {
            kc_image *inp[4] = { with(0), with(1), with(2), with(3) };
            kc_image *img = nullptr;
            s = combine_process(inp, &img);
            if (s == KC_OK) out.push_back(BandSlot{ 0, img, oy });
            break;
        }
        case KC_NODE_OUTPUT_GRAY:
        case KC_NODE_OUTPUT_RGBA: {
            if (!assigned.empty()) {
                image_retain(assigned[0].image);
                out.push_back(BandSlot{ 0, assigned[0].image, oy });
            } else {
                kc_image *img = nullptr;
                s = image_from_value(kc_size{ 1, 1 }, 0.0f, n.type == KC_NODE_OUTPUT_RGBA, &img);  // output.rs:20-31
                if (s == KC_OK) out.push_back(BandSlot{ 0, img, 0 });
            }
            break;
        }
        default: set_error("row-band evaluation: unsupported node type"); s = KC_ERR_UNSUPPORTED;
        }
    for (auto &sd : assigned) image_release(sd.image);
    return s;
}

// ---- Graph nodes ------------------------------------------------------------------------------------------
// graph::process (src/node/graph.rs:14-51) evaluates a child graph on the Graph node's inputs -- which process_node has
// resized to the node's target size T first (src/node/node_type.rs:229-237) -- and hands out the child's Output nodes as
// the Graph node's output slots.  For the band walk the child graph is spliced into a copy of the parent graph:
//   * every child node gets a fresh id, child edges keep their order;
//   * a child InputGray(i) becomes a pass-through fed by the edge that enters the Graph node on slot i; a child InputRgba by
//     the edge on the Graph node's lowest connected slot (input_rgba::process takes input_node_datas[0]).  The pass-through is
//     an Output node whose resize policy is SpecificSize(T) with the Graph node's filter: exactly the resize the Graph node
//     applied to that input (T computed here from the parents' inferred sizes by the Graph node's own policy);
//   * an edge leaving the Graph node's slot o leaves the copy of the child's Output node o instead.
// Nested Graph nodes come to the surface with their parent and are expanded in a later round.  The result holds no Graph
// node among the root's ancestors; rows, halos and arithmetic are then what the walk below does for any other graph.
int expand_graph_nodes(kc_live_graph &lg, uint32_t *root, uint32_t *slot, std::unique_ptr<kc_live_graph> &flat)
{
    auto has_graph_ancestor = [](const NodeGraph &g, uint32_t r, uint32_t *which) {
        std::vector<uint32_t> topo;
        if (topo_order(g, r, topo) != KC_OK) return false;
        for (uint32_t id : topo)
            if (g.find(id)->type == KC_NODE_GRAPH) {
                *which = id;  // topological order: the first one has no Graph node above it
                return true;
            }
        return false;
    };
    uint32_t gid = 0;
    if (!has_graph_ancestor(lg.g, *root, &gid)) return KC_OK;
    flat.reset(new kc_live_graph());
    kc_live_graph &F = *flat;
    F.tp = lg.tp;
    F.base_dir = lg.base_dir;
    F.depth = lg.depth;
    F.g = lg.g;
    for (auto &e : lg.embedded) {
        image_retain(e.image);
        F.embedded.push_back(e);
    }
    for (auto &i : lg.input_slot_datas) {
        image_retain(i.image);
        F.input_slot_datas.push_back(i);
    }
    for (int round = 0; has_graph_ancestor(F.g, *root, &gid); ++round) {
        if (round > 64) {
            set_error("Graph nodes nested too deeply");
            return KC_ERR_NODE_PROCESSING;
        }
        const Node gn = *F.g.find(gid);
        if (!gn.graph) {
            set_error("Graph node without a graph");
            return KC_ERR_NODE_PROCESSING;
        }
        // T: the Graph node's resize target, from the inferred sizes of its parents
        Walk W(F);
        KC_TRY(topo_order(F.g, gid, W.topo));
        KC_TRY(infer_sizes(W, &gid));
        const std::vector<kc_edge> in_edges = F.g.edges_into(gid);
        kc_size T{ 1, 1 };
        if (!in_edges.empty()) KC_TRY(target_size(W, gn, in_edges, &T));
        const kc_edge *lowest = nullptr;
        for (auto &e : in_edges)
            if (!lowest || e.input_slot < lowest->input_slot) lowest = &e;
        // copies of the child's nodes
        const NodeGraph &cg = *gn.graph;
        std::map<uint32_t, uint32_t> idmap;
        std::vector<kc_edge> new_edges;
        for (auto &cn : cg.nodes) {
            Node c = cn;
            c.node_id = F.g.new_id();
            idmap[cn.node_id] = c.node_id;
            if (cn.is_input()) {
                const kc_edge *feed = nullptr;
                if (cn.type == KC_NODE_INPUT_RGBA) feed = lowest;
                else
                    for (auto &e : in_edges)
                        if (e.input_slot == cn.node_id) feed = &e;
                if (!feed) {
                    set_error("row band: an Input node of a Graph node's graph has nothing connected to it");
                    return KC_ERR_NO_SLOT_DATA;
                }
                c.type = cn.type == KC_NODE_INPUT_RGBA ? KC_NODE_OUTPUT_RGBA : KC_NODE_OUTPUT_GRAY;
                c.policy = KC_POLICY_SPECIFIC_SIZE;
                c.policy_size = T;
                c.filter = gn.filter;
                c.graph.reset();
                new_edges.push_back(kc_edge{ feed->output_id, c.node_id, feed->output_slot, 0 });
            }
            F.g.nodes.push_back(c);
        }
        // edges: those into the Graph node go (their pass-through copies were made above), those out of it are re-rooted in
        // place (edge order is what the size policies see), the child's own follow
        std::vector<kc_edge> edges;
        for (auto &e : F.g.edges) {
            if (e.input_id == gid) continue;
            if (e.output_id == gid) {
                auto it = idmap.find(e.output_slot);
                if (it == idmap.end()) {
                    set_error("row band: an edge leaves a Graph node on a slot its graph has no Output node for");
                    return KC_ERR_INVALID_SLOT_ID;
                }
                edges.push_back(kc_edge{ it->second, e.input_id, 0, e.input_slot });
            } else {
                edges.push_back(e);
            }
        }
        for (auto &e : new_edges) edges.push_back(e);
        for (auto &e : cg.edges) edges.push_back(kc_edge{ idmap[e.output_id], idmap[e.input_id], e.output_slot, e.input_slot });
        F.g.edges.swap(edges);
        for (size_t i = 0; i < F.g.nodes.size(); ++i)
            if (F.g.nodes[i].node_id == gid) {
                F.g.nodes.erase(F.g.nodes.begin() + (long)i);
                break;
            }
        F.g.touch();
        if (*root == gid) {
            auto it = idmap.find(*slot);
            if (it == idmap.end()) {
                set_error("the Graph node has no such output slot");
                return KC_ERR_NO_SLOT_DATA;
            }
            *root = it->second;
            *slot = 0;
        }
    }
    F.reset_node_states();
    return KC_OK;
}

int build_walk(Walk &W, uint32_t root, int32_t y0, int32_t y1)
{
    if (!W.lg.g.find(root)) return KC_ERR_INVALID_NODE_ID;
    KC_TRY(topo_order(W.lg.g, root, W.topo));
    KC_TRY(infer_sizes(W));
    return propagate_needs(W, root, y0, y1);
}

}  // namespace

int band_source_rows(kc_live_graph &lg0, uint32_t root, int32_t y0, int32_t y1, std::vector<kc_band_rows> &out)
{
    if (!lg0.g.find(root)) return KC_ERR_INVALID_NODE_ID;
    std::unique_ptr<kc_live_graph> flat;
    uint32_t slot = 0;
    if (lg0.g.find(root)->type == KC_NODE_GRAPH) {
        // any output of the Graph node: the sources' rows are asked for per node, the first Output node stands for it
        const std::vector<uint32_t> outs = lg0.g.find(root)->graph ? lg0.g.find(root)->graph->output_ids() : std::vector<uint32_t>{};
        if (outs.empty()) return KC_ERR_NO_SLOT_DATA;
        slot = outs[0];
    }
    KC_TRY(expand_graph_nodes(lg0, &root, &slot, flat));
    kc_live_graph &lg = flat ? *flat : lg0;
    Walk W(lg);
    KC_TRY(build_walk(W, root, y0, y1));
    for (uint32_t id : W.topo) {
        const Node &n = *lg.g.find(id);
        if (!is_source(n) || !W.need[id].set) continue;
        out.push_back(kc_band_rows{ id, W.need[id].a, W.need[id].b, W.size[id].width, W.size[id].height });
    }
    return KC_OK;
}

// What a row-band PLAN needs (partition.cpp): is the graph one the band walk takes, and how large is the requested node's image?
int band_plan_info(kc_live_graph &lg0, uint32_t root, kc_size *size, bool *rgba)
{
    if (!lg0.g.find(root)) return KC_ERR_INVALID_NODE_ID;
    std::unique_ptr<kc_live_graph> flat;
    uint32_t slot = 0;
    if (lg0.g.find(root)->type == KC_NODE_GRAPH) {
        const std::vector<uint32_t> outs = lg0.g.find(root)->graph ? lg0.g.find(root)->graph->output_ids() : std::vector<uint32_t>{};
        if (outs.empty()) return KC_ERR_NO_SLOT_DATA;
        slot = outs[0];
    }
    KC_TRY(expand_graph_nodes(lg0, &root, &slot, flat));
    kc_live_graph &lg = flat ? *flat : lg0;
    Walk W(lg);
    KC_TRY(topo_order(lg.g, root, W.topo));
    KC_TRY(infer_sizes(W));
    *size = W.size[root];
    *rgba = W.rgba[root];
    return KC_OK;
}

int band_evaluate(kc_live_graph &lg0, uint32_t root, uint32_t slot, int32_t y0, int32_t y1, kc_image **out)
{
    *out = nullptr;
    KC_TRY(need_init());
    if (!lg0.g.find(root)) return KC_ERR_INVALID_NODE_ID;
    std::unique_ptr<kc_live_graph> flat;  // the graph with its Graph nodes expanded, if it has any above the root
    KC_TRY(expand_graph_nodes(lg0, &root, &slot, flat));
    kc_live_graph &lg = flat ? *flat : lg0;
    Walk W(lg);
    KC_TRY(build_walk(W, root, y0, y1));
    ResizeMemoScope memo;
    for (uint32_t id : W.topo) {
        if (!W.need[id].set) continue;
        const Node &n = *lg.g.find(id);
        if (is_source(n)) {
            KC_TRY(evaluate_source(W, n));
        } else if (n.type == KC_NODE_VALUE) {
            kc_image *img = nullptr;
            KC_TRY(value_process(n.value, &img));
            W.data[id].push_back(BandSlot{ 0, img, 0 });
        } else {
            KC_TRY(evaluate_node(W, n));
        }
    }
    const BandSlot *bs = find_band_slot(W, root, slot);
    if (!bs) {
        set_error("the node has no such slot");
        return KC_ERR_NO_SLOT_DATA;
    }
    const kc_size rs = W.size[root];
    kc_image *img = nullptr;
    KC_TRY(crop_rows(bs->img, bs->y0, rs.height, rs.height == 1 ? 0 : y0, rs.height == 1 ? 1 : y1, &img));
    int s = image_force(img);  // Clean means computed: the band is resident when this returns
    if (s != KC_OK) {
        image_release(img);
        return s;
    }
    *out = img;
    return KC_OK;
}

}  // namespace kc
