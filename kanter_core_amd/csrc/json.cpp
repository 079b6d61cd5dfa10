// NodeGraph <-> JSON in the reference's serde wire format (src/node_graph.rs:16-22,98-107;
// derive(Serialize) on Node src/node/mod.rs:113-123, NodeType src/node/node_type.rs:13-28, Edge
// src/edge.rs:8-14; sample: data/invert_graph.json).  Externally tagged enums: unit variants are
// strings ("SeparateRgba"), newtype variants one-key objects ({"Mix": "Subtract"}); NodeId / SlotId
// are bare numbers; `priority`, `cancel` and `node_id_counter` are #[serde(skip)].
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kc_runtime.hpp"

namespace kc {

namespace {

struct JVal {
    enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal *get(const char *k) const
    {
        for (auto &kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct Parser {
    const char *p, *end;
    bool ok = true;
    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool lit(const char *s)
    {
        size_t n = std::strlen(s);
        if ((size_t)(end - p) >= n && !std::memcmp(p, s, n)) {
            p += n;
            return true;
        }
        return false;
    }
    std::string string()
    {
        std::string out;
        ++p;  // opening quote
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    unsigned cp = 0;
                    for (int i = 0; i < 4 && p + 1 < end; ++i) {
                        ++p;
                        cp = cp * 16 + (unsigned)(std::isdigit((unsigned char)*p) ? *p - '0' : (std::tolower(*p) - 'a' + 10));
                    }
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 63)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 63)); out += (char)(0x80 | (cp & 63)); }
                    break;
                }
                default: out += *p; break;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) ok = false;
        else ++p;
        return out;
    }
    JVal value(int depth = 0)
    {
        JVal v;
        ws();
        if (p >= end || depth > 64) {
            ok = false;
            return v;
        }
        if (*p == '{') {
            v.t = JVal::OBJ;
            ++p;
            ws();
            if (p < end && *p == '}') { ++p; return v; }
            while (ok) {
                ws();
                if (p >= end || *p != '"') { ok = false; break; }
                std::string k = string();
                ws();
                if (p >= end || *p != ':') { ok = false; break; }
                ++p;
                v.obj.emplace_back(k, value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                ok = false;
            }
        } else if (*p == '[') {
            v.t = JVal::ARR;
            ++p;
            ws();
            if (p < end && *p == ']') { ++p; return v; }
            while (ok) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                ok = false;
            }
        } else if (*p == '"') {
            v.t = JVal::STR;
            v.str = string();
        } else if (lit("true")) {
            v.t = JVal::BOOL; v.b = true;
        } else if (lit("false")) {
            v.t = JVal::BOOL;
        } else if (lit("null")) {
            v.t = JVal::NUL;
        } else {
            char *e = nullptr;
            v.t = JVal::NUM;
            v.num = std::strtod(p, &e);
            if (e == p || e > end) ok = false;
            else p = e;
        }
        return v;
    }
};

const char *MIX_NAMES[] = { "Add", "Subtract", "Multiply", "Divide", "Pow" };
const char *FILTER_NAMES[] = { "Nearest", "Triangle", "CatmullRom", "Gaussian", "Lanczos3" };
const char *POLICY_NAMES[] = { "MostPixels", "LeastPixels", "LargestAxes", "SmallestAxes", "SpecificSlot", "SpecificSize" };
const char *TYPE_NAMES[] = { "InputGray", "InputRgba", "OutputGray", "OutputRgba", "Graph", "Image", "Embed",
                             "Write", "Value", "Mix", "HeightToNormal", "SeparateRgba", "CombineRgba" };

int find_name(const char *const *names, int n, const std::string &s)
{
    for (int i = 0; i < n; ++i)
        if (s == names[i]) return i;
    return -1;
}

bool as_u32(const JVal *v, uint32_t *out)
{
    if (!v || v->t != JVal::NUM || v->num < 0 || v->num > 4294967295.0 || v->num != std::floor(v->num)) return false;
    *out = (uint32_t)v->num;
    return true;
}

bool graph_from_jval(const JVal &root, NodeGraph &g, int depth);

bool node_from_jval(const JVal &j, Node &n, int depth)
{
    if (j.t != JVal::OBJ || !as_u32(j.get("node_id"), &n.node_id)) return false;
    const JVal *nt = j.get("node_type");
    if (!nt) return false;
    std::string tag;
    const JVal *payload = nullptr;
    if (nt->t == JVal::STR) {
        tag = nt->str;
    } else if (nt->t == JVal::OBJ && nt->obj.size() == 1) {
        tag = nt->obj[0].first;
        payload = &nt->obj[0].second;
    } else {
        return false;
    }
    n.type = find_name(TYPE_NAMES, 13, tag);
    switch (n.type) {
    case KC_NODE_INPUT_GRAY: case KC_NODE_INPUT_RGBA: case KC_NODE_OUTPUT_GRAY: case KC_NODE_OUTPUT_RGBA:
    case KC_NODE_IMAGE: case KC_NODE_WRITE:
        if (!payload || payload->t != JVal::STR) return false;
        n.text = payload->str;
        break;
    case KC_NODE_GRAPH:
        if (!payload || depth > 16) return false;
        n.graph = std::make_shared<NodeGraph>();
        if (!graph_from_jval(*payload, *n.graph, depth + 1)) return false;
        break;
    case KC_NODE_EMBED:
        if (!as_u32(payload, &n.embed_id)) return false;
        break;
    case KC_NODE_VALUE:
        if (!payload || payload->t != JVal::NUM) return false;
        n.value = (float)payload->num;
        break;
    case KC_NODE_MIX:
        if (!payload || payload->t != JVal::STR) return false;
        n.mix_type = find_name(MIX_NAMES, 5, payload->str);
        if (n.mix_type < 0) return false;
        break;
    case KC_NODE_HEIGHT_TO_NORMAL: case KC_NODE_SEPARATE_RGBA: case KC_NODE_COMBINE_RGBA:
        break;
    default:
        return false;
    }
    const JVal *rp = j.get("resize_policy");
    if (!rp) return false;  // serde: missing field is an error
    if (rp->t == JVal::STR) {
        n.policy = find_name(POLICY_NAMES, 4, rp->str);
        if (n.policy < 0) return false;
    } else if (rp->t == JVal::OBJ && rp->obj.size() == 1) {
        const std::string &k = rp->obj[0].first;
        const JVal &v = rp->obj[0].second;
        if (k == "SpecificSlot") {
            n.policy = KC_POLICY_SPECIFIC_SLOT;
            if (!as_u32(&v, &n.policy_slot)) return false;
        } else if (k == "SpecificSize") {
            n.policy = KC_POLICY_SPECIFIC_SIZE;
            if (v.t != JVal::OBJ || !as_u32(v.get("width"), &n.policy_size.width) ||
                !as_u32(v.get("height"), &n.policy_size.height))
                return false;
        } else {
            return false;
        }
    } else {
        return false;
    }
    const JVal *rf = j.get("resize_filter");
    if (!rf || rf->t != JVal::STR) return false;
    n.filter = find_name(FILTER_NAMES, 5, rf->str);
    return n.filter >= 0;
}

bool graph_from_jval(const JVal &root, NodeGraph &g, int depth)
{
    if (root.t != JVal::OBJ) return false;
    const JVal *nodes = root.get("nodes"), *edges = root.get("edges");
    if (!nodes || nodes->t != JVal::ARR || !edges || edges->t != JVal::ARR) return false;
    for (auto &jn : nodes->arr) {
        Node n;
        if (!node_from_jval(jn, n, depth)) return false;
        g.nodes.push_back(std::move(n));
        g.touch();
    }
    for (auto &je : edges->arr) {
        kc_edge e;
        if (je.t != JVal::OBJ || !as_u32(je.get("output_id"), &e.output_id) || !as_u32(je.get("input_id"), &e.input_id) ||
            !as_u32(je.get("output_slot"), &e.output_slot) || !as_u32(je.get("input_slot"), &e.input_slot))
            return false;
        g.edges.push_back(e);
        g.touch();
    }
    // NodeGraph::from_path, src/node_graph.rs:36-43: counter = max id + 1
    uint32_t mx = 0;
    bool any = false;
    for (auto &n : g.nodes) {
        if (!any || n.node_id > mx) mx = n.node_id;
        any = true;
    }
    g.node_id_counter = any ? mx + 1 : 0;
    return true;
}

void esc(std::string &o, const std::string &s)
{
    o += '"';
    for (unsigned char ch : s) {
        switch (ch) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\n': o += "\\n"; break;
        case '\t': o += "\\t"; break;
        case '\r': o += "\\r"; break;
        default:
            if (ch < 0x20) {
                char b[8];
                std::snprintf(b, sizeof(b), "\\u%04x", ch);
                o += b;
            } else {
                o += (char)ch;
            }
        }
    }
    o += '"';
}

// Shortest decimal that round-trips the f32 (what serde_json's ryu prints), always with a
// fractional part or exponent so it reads back as a float.
std::string f32_repr(float v)
{
    if (std::isnan(v) || std::isinf(v)) return "null";  // serde_json writes null for non-finite floats
    char b[64];
    for (int prec = 1; prec <= 9; ++prec) {
        std::snprintf(b, sizeof(b), "%.*g", prec, (double)v);
        if (std::strtof(b, nullptr) == v) break;
    }
    std::string s = b;
    if (s.find('.') == std::string::npos && s.find('e') == std::string::npos && s.find("inf") == std::string::npos) s += ".0";
    return s;
}

void ind(std::string &o, int n) { o.append((size_t)n * 2, ' '); }

void graph_write(std::string &o, const NodeGraph &g, int lvl);

void node_write(std::string &o, const Node &n, int lvl)
{
    ind(o, lvl); o += "{\n";
    ind(o, lvl + 1); o += "\"node_id\": " + std::to_string(n.node_id) + ",\n";
    ind(o, lvl + 1); o += "\"node_type\": ";
    const char *tag = TYPE_NAMES[n.type];
    switch (n.type) {
    case KC_NODE_HEIGHT_TO_NORMAL: case KC_NODE_SEPARATE_RGBA: case KC_NODE_COMBINE_RGBA:
        o += std::string("\"") + tag + "\"";
        break;
    default:
        o += "{\n";
        ind(o, lvl + 2); o += std::string("\"") + tag + "\": ";
        switch (n.type) {
        case KC_NODE_GRAPH: {
            NodeGraph empty;
            graph_write(o, n.graph ? *n.graph : empty, lvl + 2);
            break;
        }
        case KC_NODE_EMBED: o += std::to_string(n.embed_id); break;
        case KC_NODE_VALUE: o += f32_repr(n.value); break;
        case KC_NODE_MIX: esc(o, MIX_NAMES[n.mix_type]); break;
        default: esc(o, n.text); break;
        }
        o += "\n";
        ind(o, lvl + 1); o += "}";
    }
    o += ",\n";
    ind(o, lvl + 1); o += "\"resize_policy\": ";
    if (n.policy == KC_POLICY_SPECIFIC_SLOT) {
        o += "{\n";
        ind(o, lvl + 2); o += "\"SpecificSlot\": " + std::to_string(n.policy_slot) + "\n";
        ind(o, lvl + 1); o += "}";
    } else if (n.policy == KC_POLICY_SPECIFIC_SIZE) {
        o += "{\n";
        ind(o, lvl + 2); o += "\"SpecificSize\": {\n";
        ind(o, lvl + 3); o += "\"width\": " + std::to_string(n.policy_size.width) + ",\n";
        ind(o, lvl + 3); o += "\"height\": " + std::to_string(n.policy_size.height) + "\n";
        ind(o, lvl + 2); o += "}\n";
        ind(o, lvl + 1); o += "}";
    } else {
        esc(o, POLICY_NAMES[n.policy]);
    }
    o += ",\n";
    ind(o, lvl + 1); o += "\"resize_filter\": ";
    esc(o, FILTER_NAMES[n.filter]);
    o += "\n";
    ind(o, lvl); o += "}";
}

void graph_write(std::string &o, const NodeGraph &g, int lvl)
{
    o += "{\n";
    ind(o, lvl + 1); o += "\"nodes\": [";
    for (size_t i = 0; i < g.nodes.size(); ++i) {
        o += i ? ",\n" : "\n";
        node_write(o, g.nodes[i], lvl + 2);
    }
    if (!g.nodes.empty()) { o += "\n"; ind(o, lvl + 1); }
    o += "],\n";
    ind(o, lvl + 1); o += "\"edges\": [";
    for (size_t i = 0; i < g.edges.size(); ++i) {
        const kc_edge &e = g.edges[i];
        o += i ? ",\n" : "\n";
        ind(o, lvl + 2); o += "{\n";
        ind(o, lvl + 3); o += "\"output_id\": " + std::to_string(e.output_id) + ",\n";
        ind(o, lvl + 3); o += "\"input_id\": " + std::to_string(e.input_id) + ",\n";
        ind(o, lvl + 3); o += "\"output_slot\": " + std::to_string(e.output_slot) + ",\n";
        ind(o, lvl + 3); o += "\"input_slot\": " + std::to_string(e.input_slot) + "\n";
        ind(o, lvl + 2); o += "}";
    }
    if (!g.edges.empty()) { o += "\n"; ind(o, lvl + 1); }
    o += "]\n";
    ind(o, lvl); o += "}";
}

}  // namespace

int graph_from_json(const std::string &text, NodeGraph &g)
{
    Parser ps{ text.data(), text.data() + text.size() };
    JVal root = ps.value();
    ps.ws();
    if (!ps.ok || ps.p != ps.end) {
        set_error("malformed JSON");
        return KC_ERR_IO;
    }
    NodeGraph tmp;
    if (!graph_from_jval(root, tmp, 0)) {
        set_error("JSON does not describe a NodeGraph");
        return KC_ERR_IO;
    }
    g = std::move(tmp);
    return KC_OK;
}

std::string graph_to_json(const NodeGraph &g)
{
    std::string o;
    graph_write(o, g, 0);
    return o;
}

}  // namespace kc
