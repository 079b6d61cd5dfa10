"""The graphs of the reference's golden-image tests (tests/integration_tests.rs), written as the
reference's own serde JSON shape (src/node_graph.rs:16-22, data/invert_graph.json) so that both
the CPU oracle (oracle.RefGraph) and the HIP product (NodeGraph.from_json) can evaluate them.
Each entry: name -> (graph dict, node to read, golden PNG under tests/golden/test_compare)."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INPUTS = os.path.join(GOLDEN, "inputs")
COMPARE = os.path.join(GOLDEN, "test_compare")

IMAGE_1, IMAGE_2 = "image_1.png", "image_2.png"
HEART_110, HEART_128, HEART_256 = "heart_110.png", "heart_128.png", "heart_256.png"
HEART_WIDE, HEART_TALL, CLOUDS = "heart_wide.png", "heart_tall.png", "clouds.png"


class G:
    """Tiny NodeGraph builder: ids are handed out 0, 1, 2 ... like NodeGraph::new_id
    (src/node_graph.rs:86-96); edges keep insertion order and `connect` replaces an occupied
    input slot (src/node_graph.rs:416-446)."""

    def __init__(self):
        self.nodes, self.edges = [], []

    def add(self, node_type, policy="MostPixels", filt="Triangle"):
        nid = len(self.nodes)
        self.nodes.append({"node_id": nid, "node_type": node_type, "resize_policy": policy, "resize_filter": filt})
        return nid

    def connect(self, out_id, in_id, out_slot, in_slot):
        self.edges = [e for e in self.edges if not (e["input_id"] == in_id and e["input_slot"] == in_slot)]
        self.edges.append({"output_id": out_id, "input_id": in_id, "output_slot": out_slot, "input_slot": in_slot})

    def dict(self):
        return {"nodes": self.nodes, "edges": self.edges}


def _invert_graph():
    # tests/integration_tests.rs:995-1022
    g = G()
    white = g.add({"Value": 1.0})
    inp = g.add({"InputGray": "in"})
    sub = g.add({"Mix": "Subtract"})
    out = g.add({"OutputGray": "out"})
    g.connect(white, sub, 0, 0)
    g.connect(inp, sub, 0, 1)
    g.connect(sub, out, 0, 0)
    return g.dict(), inp, out


def mix_gray(op):
    # mix_node_test_gray, tests/integration_tests.rs:1439-1475
    g = G()
    img = g.add({"Image": IMAGE_2})
    sep = g.add("SeparateRgba")
    mix = g.add({"Mix": op})
    out = g.add({"OutputGray": "out"})
    g.connect(img, sep, 0, 0)
    g.connect(sep, mix, 0, 0)
    g.connect(sep, mix, 1, 1)
    g.connect(mix, out, 0, 0)
    return g.dict(), out


def mix_rgba(op):
    # mix_node_test_rgba, tests/integration_tests.rs:1477-1510
    g = G()
    i1 = g.add({"Image": IMAGE_1})
    i2 = g.add({"Image": IMAGE_2})
    mix = g.add({"Mix": op})
    out = g.add({"OutputRgba": "out"})
    g.connect(i1, mix, 0, 0)
    g.connect(i2, mix, 0, 1)
    g.connect(mix, out, 0, 0)
    return g.dict(), out


def mix_single_input(op, slot):
    # tests/integration_tests.rs:496-553
    g = G()
    img = g.add({"Image": IMAGE_2})
    mix = g.add({"Mix": op})
    out = g.add({"OutputGray": "out"})
    g.connect(img, mix, 0, slot)
    g.connect(mix, out, 0, 0)
    return g.dict(), out


def separate_node():
    # tests/integration_tests.rs:621-674
    g = G()
    i1 = g.add({"Image": IMAGE_1})
    s1 = g.add("SeparateRgba")
    i2 = g.add({"Image": IMAGE_2})
    s2 = g.add("SeparateRgba")
    out = g.add({"OutputRgba": "out"})
    comb = g.add("CombineRgba")
    g.connect(i1, s1, 0, 0)
    g.connect(i2, s2, 0, 0)
    g.connect(s1, comb, 3, 0)
    g.connect(s1, comb, 1, 1)
    g.connect(s2, comb, 2, 2)
    g.connect(s2, comb, 3, 3)
    g.connect(comb, out, 0, 0)
    return g.dict(), out


def irregular_sizes():
    # tests/integration_tests.rs:678-738
    g = G()
    i1 = g.add({"Image": HEART_128})
    i2 = g.add({"Image": HEART_110})
    mix = g.add({"Mix": "Add"})
    out = g.add({"OutputRgba": "out"})
    g.connect(i1, mix, 0, 0)
    g.connect(i2, mix, 0, 1)
    g.connect(mix, out, 0, 0)
    return g.dict(), out


def value_node():
    # tests/integration_tests.rs:814-846
    g = G()
    vals = [g.add({"Value": v}) for v in (0.0, 0.33, 0.66, 1.0)]
    comb = g.add("CombineRgba", policy={"SpecificSize": {"width": 256, "height": 256}})
    for i, v in enumerate(vals):
        g.connect(v, comb, 0, i)
    return g.dict(), comb


def invert_graph_node(imported=False):
    # tests/integration_tests.rs:993-1071 and :1110-1160
    if imported:
        with open(os.path.join(INPUTS, "invert_graph.json")) as f:
            inner = json.load(f)
        inp = next(n["node_id"] for n in inner["nodes"] if n["node_type"] == {"InputGray": "in"})
        outn = next(n["node_id"] for n in inner["nodes"] if n["node_type"] == {"OutputGray": "out"})
    else:
        inner, inp, outn = _invert_graph()
    g = G()
    img = g.add({"Image": IMAGE_2})
    if imported:
        sep = g.add("SeparateRgba")
        gn = g.add({"Graph": inner})
    else:
        gn = g.add({"Graph": inner})
        sep = g.add("SeparateRgba")
    out = g.add({"OutputGray": "out"})
    g.connect(img, sep, 0, 0)
    g.connect(sep, gn, 0, inp)
    g.connect(gn, out, outn, 0)
    return g.dict(), out


def graph_node(rgba):
    # tests/integration_tests.rs:1209-1328
    inner = G()
    i = inner.add({"InputRgba" if rgba else "InputGray": "in"})
    o = inner.add({"OutputRgba" if rgba else "OutputGray": "out"})
    inner.connect(i, o, 0, 0)
    g = G()
    img = g.add({"Image": IMAGE_2})
    if rgba:
        gn = g.add({"Graph": inner.dict()})
        out = g.add({"OutputRgba": "out"})
        g.connect(img, gn, 0, i)
    else:
        sep = g.add("SeparateRgba")
        gn = g.add({"Graph": inner.dict()})
        out = g.add({"OutputGray": "out"})
        g.connect(img, sep, 0, 0)
        g.connect(sep, gn, 0, i)
    g.connect(gn, out, o, 0)
    return g.dict(), out


def input_output(path=IMAGE_2):
    # tests/integration_tests.rs:53-95
    g = G()
    img = g.add({"Image": path})
    out = g.add({"OutputRgba": "out"})
    g.connect(img, out, 0, 0)
    return g.dict(), out


def height_to_normal_node():
    # tests/integration_tests.rs:1351-1384
    g = G()
    img = g.add({"Image": CLOUDS})
    sep = g.add("SeparateRgba")
    h2n = g.add("HeightToNormal")
    out = g.add({"OutputRgba": "out"})
    g.connect(img, sep, 0, 0)
    g.connect(sep, h2n, 0, 0)
    g.connect(h2n, out, 0, 0)
    return g.dict(), out


def resize_policy(policy, img1, img2):
    # resize_policy_test, tests/integration_tests.rs:848-892
    g = G()
    i1 = g.add({"Image": img1})
    i2 = g.add({"Image": img2})
    mix = g.add({"Mix": "Add"}, policy=policy)
    g.connect(i1, mix, 0, 0)
    g.connect(i2, mix, 0, 1)
    return g.dict(), mix


GOLDEN_CASES = {}
for _op, _name in (("Add", "add"), ("Subtract", "subtract"), ("Multiply", "multiply"), ("Divide", "divide"),
                   ("Pow", "pow")):
    GOLDEN_CASES[_name + "_node_gray"] = mix_gray(_op) + (_name + "_node_gray.png",)
    GOLDEN_CASES[_name + "_node_rgba"] = mix_rgba(_op) + (_name + "_node_rgba.png",)
GOLDEN_CASES["mix_node_single_input"] = mix_single_input("Add", 0) + ("mix_node_single_input.png",)
GOLDEN_CASES["mix_node_single_input_2"] = mix_single_input("Subtract", 1) + ("mix_node_single_input_2.png",)
GOLDEN_CASES["separate_node"] = separate_node() + ("mix_images.png",)
GOLDEN_CASES["irregular_sizes"] = irregular_sizes() + ("irregular_sizes.png",)
GOLDEN_CASES["value_node"] = value_node() + ("value_node.png",)
GOLDEN_CASES["invert_graph_node"] = invert_graph_node(False) + ("invert_graph_node.png",)
GOLDEN_CASES["invert_graph_node_import"] = invert_graph_node(True) + ("invert_graph_node_import.png",)
GOLDEN_CASES["graph_node_rgba"] = graph_node(True) + ("graph_node_rgba.png",)
GOLDEN_CASES["graph_node_gray"] = graph_node(False) + ("graph_node_gray.png",)
GOLDEN_CASES["input_output"] = input_output() + ("input_output.png",)
GOLDEN_CASES["embedded_node_data"] = input_output(IMAGE_1) + ("embedded_node_data.png",)
GOLDEN_CASES["height_to_normal_node"] = height_to_normal_node() + ("height_to_normal_node.png",)

# Size-only cases, tests/integration_tests.rs:894-949: (policy, image 1, image 2, expected size)
RESIZE_POLICY_CASES = [
    ("LeastPixels", HEART_128, HEART_256, (128, 128)),
    ("LargestAxes", HEART_WIDE, HEART_TALL, (128, 128)),
    ("SmallestAxes", HEART_WIDE, HEART_TALL, (64, 64)),
    ("MostPixels", HEART_128, HEART_256, (256, 256)),
    ({"SpecificSize": {"width": 256, "height": 256}}, HEART_128, HEART_WIDE, (256, 256)),
    ({"SpecificSlot": 1}, HEART_128, HEART_WIDE, (128, 64)),
    ({"SpecificSlot": 2}, HEART_128, HEART_WIDE, (128, 128)),
]
