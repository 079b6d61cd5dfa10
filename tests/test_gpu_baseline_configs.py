"""GPU parity for the BASELINE.json configs that had no test of their own:
  #0  data/invert_graph.json on data/heart_256.png, exactly as BASELINE states it;
  #4  8 independent 16-node subgraphs + the fixed-order 7-node Mix(Add) tree at 4096x4096 (on one GPU);
  #1  with Mix(Pow) at 4096x4096 (<= 1 ulp: powf is libm's in the reference, src/node/mix.rs:189).
Inputs follow SURVEY.md 8(d): splitmix planes, per-subgraph seeds 0x5EED0100 + k / 0x5EED0200 + k."""
import json
import os

import numpy as np
import pytest

from golden_graphs import G, HEART_256, INPUTS
from pngio import read_png
from util import SEED_A, SEED_B, assert_planes, bit_equal, splitmix_plane

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    return kc


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    orc.set_threads(8)
    yield orc
    orc.set_threads(1)


# ------------------------------------------------------------------------------------ config #0
def _config0_graph():
    with open(os.path.join(INPUTS, "invert_graph.json")) as f:
        inner = json.load(f)
    inp = next(n["node_id"] for n in inner["nodes"] if n["node_type"] == {"InputGray": "in"})
    outn = next(n["node_id"] for n in inner["nodes"] if n["node_type"] == {"OutputGray": "out"})
    g = G()
    img = g.add({"Image": HEART_256})
    sep = g.add("SeparateRgba")
    gn = g.add({"Graph": inner})
    out = g.add({"OutputGray": "out"})
    g.connect(img, sep, 0, 0)
    g.connect(sep, gn, 0, inp)
    g.connect(gn, out, outn, 0)
    return g.dict(), out


def test_config0_invert_graph_json_on_heart_256(kc, orc, load_image):
    """tests/integration_tests.rs:1110-1160 with data/heart_256.png as the image: the nested graph is read
    from the reference's own JSON file by NodeGraph.from_path; no reference golden exists for this input
    (SURVEY 8(d)), so the oracle's literal process_node restatement and the closed form 1 - R/255 pin it."""
    graph, out = _config0_graph()
    ref = orc.RefGraph(graph, load_image)
    want_f32 = ref.slot_data(out, 0).image.planes
    want_u8 = ref.buffer_rgba(out, 0)

    inner = kc.NodeGraph.from_path(os.path.join(INPUTS, "invert_graph.json"))
    lg = kc.TextureProcessor.new(10_000_000).new_live_graph()
    lg.set_base_dir(INPUTS)
    image_node = lg.add_node(kc.Node.new(kc.NodeType.Image(HEART_256)))
    separate_node = lg.add_node(kc.Node.new(kc.NodeType.SeparateRgba))
    graph_node = lg.add_node(kc.Node.new(kc.NodeType.Graph(inner)))
    output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputGray("out")))
    lg.connect(image_node, separate_node, 0, 0)
    lg.connect(separate_node, graph_node, 0, inner.input_slot_id_with_name("in"))
    lg.connect(graph_node, output_node, inner.output_slot_id_with_name("out"), 0)
    kc.LiveGraph.await_clean_read(lg, output_node)
    got_u8 = lg.buffer_rgba(output_node, kc.SlotId(0))
    got = lg.slot_data(output_node, 0).image
    assert not got.is_rgba() and lg.slot_data_size(output_node, 0) == (256, 256)
    assert_planes(got.planes(), want_f32, what="config #0 f32")
    assert np.array_equal(got_u8, want_u8)
    # closed form: gray(1 - R/255), truncated
    r = read_png(os.path.join(INPUTS, HEART_256))[..., 0].astype(np.float32) / np.float32(255.0)
    inv = np.float32(1.0) - r
    assert bit_equal(got.planes()[0], inv)
    q = np.minimum(np.clip(inv, 0, 1) * np.float32(255.0), np.float32(255.0)).astype(np.uint8)
    assert np.array_equal(got_u8[..., 0], q) and np.array_equal(got_u8[..., 3], np.full_like(q, 255))


# ------------------------------------------------------------------------------------ config #4
def _add_chain(kc, lg, src_a, src_b, n_nodes):
    one = lg.add_node(kc.Node.new(kc.NodeType.Value(1.0)))
    white = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
    for s in range(3):
        lg.connect(one, white, 0, s)
    prev = src_a
    for i in range(1, n_nodes + 1):
        if i & 1:
            n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply if (i >> 1) & 1 else kc.MixType.Add)))
            lg.connect(prev, n, 0, 0)
            lg.connect(src_b, n, 0, 1)
        else:
            n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
            lg.connect(white, n, 0, 0)
            lg.connect(prev, n, 0, 1)
        prev = n
    return prev


def _tree(items, combine):
    items = list(items)
    while len(items) > 1:
        nxt = [combine(items[i], items[i + 1]) for i in range(0, len(items) - 1, 2)]
        if len(items) & 1:
            nxt.append(items[-1])
        items = nxt
    return items[0]


@pytest.mark.parametrize("use_cache", [False, True])
def test_config4_fanin_8x16_nodes_4096_vs_oracle(kc, orc, use_cache):
    """BASELINE config #4 as ONE graph on one GPU: 8 independent 16-node subgraphs (SURVEY 8(d) config #3's
    chain, seeds per subgraph) whose results are summed by a 7-node Mix(Add) tree with the fixed pairing
    ((0,1),(2,3)),((4,5),(6,7)).  Every plane of the result bit-equal to the oracle's node-by-node
    evaluation; use_cache = True materialises all 135 nodes."""
    S, B, N = 4096, 8, 16
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    lg.use_cache = use_cache
    lasts, want = [], []
    for k in range(B):
        a = [splitmix_plane(0x5EED0100 + k, c, S, S) for c in range(4)]
        b = [splitmix_plane(0x5EED0200 + k, c, S, S) for c in range(4)]
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 2 * k)
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(b)), 2 * k + 1)
        na = lg.add_node(kc.Node.new(kc.NodeType.Embed(2 * k)))
        nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(2 * k + 1)))
        lasts.append(_add_chain(kc, lg, na, nb, N))
        want.append(orc.chain32(a, b, N)[:3])
        del a, b

    def add(x, y):
        n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
        lg.connect(x, n, 0, 0)
        lg.connect(y, n, 0, 1)
        return n

    root = _tree(lasts, add)
    launches0 = kc.stats()["kernel_launches"]
    got = lg.await_clean(root).slot_data(root, 0).image
    launches = kc.stats()["kernel_launches"] - launches0
    assert got.is_rgba() and lg.slot_data_size(root, 0) == (S, S)
    total = _tree(want, lambda x, y: [orc.mix_plane("Add", x[c], y[c]) for c in range(3)])
    assert_planes(got.planes(), total + [np.ones((S, S), np.float32)], what="config #4 use_cache=%s" % use_cache)
    if not use_cache:
        assert launches <= 24, launches  # 8 fused subgraphs + the pieces of the add tree, not 135 nodes


# ------------------------------------------------------------------------------------ config #1, Pow
def test_config1_mix_pow_rgba_4096_within_one_ulp(kc, orc):
    S = 4096
    a = [splitmix_plane(SEED_A, c, S, S) for c in range(3)] + [np.ones((S, S), np.float32)]
    b = [splitmix_plane(SEED_B, c, S, S) for c in range(3)] + [np.ones((S, S), np.float32)]
    got = kc.mix_process(kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b), kc.MixType.Pow).planes()
    want = [orc.mix_plane("Pow", a[c], b[c]) for c in range(3)] + [np.ones((S, S), np.float32)]
    assert_planes(got[:3], want[:3], ulp=1, what="4096 Pow")
    assert bit_equal(got[3], want[3])
    # and the fraction that is not bit-identical stays tiny (f64-rounded-once vs glibc powf)
    diff = sum(int((g.view(np.uint32) != w.view(np.uint32)).sum()) for g, w in zip(got[:3], want[:3]))
    assert diff < 3 * S * S // 1000, diff
