"""The reference's integration tests (tests/integration_tests.rs) rewritten against the mirror API
(TextureProcessor / LiveGraph / Node ...), on the HIP backend.  Each test names the reference
test it follows; golden PNGs are the reference's own (tests/golden/test_compare)."""
import os

import numpy as np
import pytest

from golden_graphs import COMPARE, INPUTS
from pngio import read_png

pytestmark = pytest.mark.gpu

IMAGE_1, IMAGE_2 = "image_1.png", "image_2.png"
HEART_128, HEART_110, CLOUDS = "heart_128.png", "heart_110.png", "clouds.png"


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    return kc


def tex_pro_new(kc):
    return kc.TextureProcessor.new(10_000_000)


def new_graph(kc):
    tex_pro = tex_pro_new(kc)
    live_graph = tex_pro.new_live_graph()
    live_graph.set_base_dir(INPUTS)
    return tex_pro, live_graph


def images_equal(buf, name):
    want = read_png(os.path.join(COMPARE, name))
    return buf.shape == want.shape and np.array_equal(buf, want)


def save_and_compare(kc, live_graph, node_id, name):
    buf = kc.LiveGraph.await_clean_read(live_graph, node_id).buffer_rgba(node_id, kc.SlotId(0))
    assert images_equal(buf, name), name


def test_input_output(kc):  # :53-95
    _, lg = new_graph(kc)
    input_node = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_2)))
    output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
    lg.connect(input_node, output_node, kc.SlotId(0), kc.SlotId(0))
    buf = kc.LiveGraph.await_clean_read(lg, output_node).buffer_rgba(output_node, kc.SlotId(0))
    assert np.array_equal(buf, read_png(os.path.join(INPUTS, IMAGE_2)))


def test_deadlock(kc):  # :109-138: one output feeding both Mix slots
    _, lg = new_graph(kc)
    value_node = lg.add_node(kc.Node.new(kc.NodeType.Value(0.0)))
    mix_node_1 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
    lg.connect(value_node, mix_node_1, 0, 0)
    lg.connect(value_node, mix_node_1, 0, 1)
    assert kc.LiveGraph.await_clean_read(lg, mix_node_1).slot_data(mix_node_1, 0).size() == (1, 1)


def test_drive_cache_values_survive(kc):  # :140-247 (the spill assertions have no HBM analogue)
    VAL = [0.0, 0.3, 0.7, 1.0]
    tex_pro = tex_pro_new(kc)
    tex_pro.memory_threshold = 16
    lg = tex_pro.new_live_graph()
    lg.use_cache = True
    rgba_node = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
    value_nodes = []
    for i, val in enumerate(VAL):
        n = lg.add_node(kc.Node.new(kc.NodeType.Value(val)))
        value_nodes.append(n)
        lg.connect(n, rgba_node, 0, i)
    mix_node_1 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
    mix_node_2 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
    lg.connect(rgba_node, mix_node_1, 0, 0)
    lg.connect(mix_node_1, mix_node_2, 0, 0)
    kc.LiveGraph.await_clean_read(lg, mix_node_2)
    for n in value_nodes + [rgba_node, mix_node_1, mix_node_2]:
        assert lg.slot_in_memory(n, 0)
    pixel = [float(p[0, 0]) for p in lg.slot_data(rgba_node, 0).image.planes()]
    assert pixel == [float(np.float32(v)) for v in VAL]


def test_no_cache_and_use_cache(kc):  # :249-305
    for use_cache in (False, True):
        _, lg = new_graph(kc)
        lg.use_cache = use_cache
        value_node = lg.add_node(kc.Node.new(kc.NodeType.Value(1.0)))
        output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputGray("out")))
        lg.connect(value_node, output_node, 0, 0)
        g = kc.LiveGraph.await_clean_read(lg, output_node)
        if use_cache:
            g.slot_data(value_node, 0)
        else:
            with pytest.raises(kc.TexProError):
                g.slot_data(value_node, 0)


def test_request_empty_buffer(kc):  # :307-333
    _, lg = new_graph(kc)
    mix_node = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.default())))
    output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
    lg.connect(mix_node, output_node, 0, 0)
    buf = kc.LiveGraph.await_clean_read(lg, output_node).buffer_rgba(output_node, 0)
    assert buf.reshape(-1).tolist() == [0, 0, 0, 255]


def test_input_output_intercept_lanczos_chain(kc):  # :335-410: 256 -> 10 -> 20 -> 30, Lanczos3
    from oracle import oracle as orc
    _, lg = new_graph(kc)
    lg.auto_update = True
    input_node = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_2)))
    prev = input_node
    nodes = []
    for size in (10, 20, 30):
        n = kc.Node.new(kc.NodeType.Mix(kc.MixType.default()))
        n.resize_filter = kc.ResizeFilter.Lanczos3
        n.resize_policy = kc.ResizePolicy.SpecificSize(kc.Size(size, size))
        nid = lg.add_node(n)
        lg.connect(prev, nid, 0, 0)
        nodes.append(nid)
        prev = nid
    output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
    lg.connect(prev, output_node, 0, 0)
    g = kc.LiveGraph.await_clean_read(lg, output_node)
    assert g.node_state(nodes[0]) == kc.NodeState.Clean and g.slot_data_size(output_node, 0) == (30, 30)
    # same chain on the oracle: resize -> Mix(Add)(x, zeros) three times
    planes = orc.deconstruct_u8(read_png(os.path.join(INPUTS, IMAGE_2)))
    for size in (10, 20, 30):
        planes = [orc.resize_plane(p, size, size, "Lanczos3") for p in planes]
        planes = [orc.mix_plane("Add", p, np.zeros_like(p)) for p in planes[:3]] + [np.ones_like(planes[0])]
    got = g.slot_data(output_node, 0).image.planes()
    for a, b in zip(got, planes):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("mix,slot,golden", [("Add", 0, "mix_node_single_input.png"),
                                             ("Subtract", 1, "mix_node_single_input_2.png")])
def test_mix_node_single_input(kc, mix, slot, golden):  # :494-553
    _, lg = new_graph(kc)
    value_node = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_2)))
    mix_node = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.parse(mix))))
    output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputGray("out")))
    lg.connect(value_node, mix_node, 0, slot)
    lg.connect(mix_node, output_node, 0, 0)
    save_and_compare(kc, lg, output_node, golden)


def test_embedded_node_data(kc):  # :567-617
    tex_pro = tex_pro_new(kc)
    lg_embed = tex_pro.new_live_graph()
    lg_embed.set_base_dir(INPUTS)
    input_node = lg_embed.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_1)))
    output_node = lg_embed.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
    lg_embed.connect(input_node, output_node, 0, 0)
    slot_data = kc.LiveGraph.await_clean_read(lg_embed, output_node).slot_data(output_node, 0)
    lg_out = tex_pro.new_live_graph()
    out2 = lg_out.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
    esd_id = lg_out.embed_slot_data_with_id(slot_data, kc.EmbeddedSlotDataId(0))
    inp = lg_out.add_node(kc.Node.new(kc.NodeType.Embed(esd_id)))
    lg_out.connect(inp, out2, 0, 0)
    save_and_compare(kc, lg_out, out2, "embedded_node_data.png")
    with pytest.raises(kc.TexProError):
        lg_out.embed_slot_data_with_id(slot_data, kc.EmbeddedSlotDataId(0))


def test_separate_node(kc):  # :619-674
    _, lg = new_graph(kc)
    input_1 = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_1)))
    separate_1 = lg.add_node(kc.Node.new(kc.NodeType.SeparateRgba))
    input_2 = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_2)))
    separate_2 = lg.add_node(kc.Node.new(kc.NodeType.SeparateRgba))
    output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
    combine = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
    lg.connect(input_1, separate_1, 0, 0)
    lg.connect(input_2, separate_2, 0, 0)
    lg.connect(separate_1, combine, 3, 0)
    lg.connect(separate_1, combine, 1, 1)
    lg.connect(separate_2, combine, 2, 2)
    lg.connect(separate_2, combine, 3, 3)
    lg.connect(combine, output_node, 0, 0)
    save_and_compare(kc, lg, output_node, "mix_images.png")


def test_irregular_sizes(kc):  # :676-738: Triangle 110 -> 128 inside Mix
    _, lg = new_graph(kc)
    input_1 = lg.add_node(kc.Node.new(kc.NodeType.Image(HEART_128)))
    input_2 = lg.add_node(kc.Node.new(kc.NodeType.Image(HEART_110)))
    mix = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.default())))
    output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
    lg.connect(input_1, mix, 0, 0)
    lg.connect(input_2, mix, 0, 1)
    lg.connect(mix, output_node, 0, 0)
    size = kc.LiveGraph.await_clean_read(lg, output_node).slot_data_size(output_node, 0)
    assert size == (128, 128)
    save_and_compare(kc, lg, output_node, "irregular_sizes.png")


def test_unconnected_node(kc):  # :740-770
    _, lg = new_graph(kc)
    input_1 = lg.add_node(kc.Node.new(kc.NodeType.Value(0.0)))
    lg.add_node(kc.Node.new(kc.NodeType.Value(0.0)))
    output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputGray("out")))
    lg.connect(input_1, output_node, 0, 0)
    lg.auto_update = True
    assert kc.LiveGraph.await_clean_read(lg, output_node).buffer_rgba(output_node, 0).reshape(-1).tolist() == [0, 0, 0, 255]


def test_value_node(kc):  # :812-846
    _, lg = new_graph(kc)
    ids = [lg.add_node(kc.Node.new(kc.NodeType.Value(v))) for v in (0.0, 0.33, 0.66, 1.0)]
    node = kc.Node.new(kc.NodeType.CombineRgba)
    node.resize_policy = kc.ResizePolicy.SpecificSize(kc.Size(256, 256))
    combine_node = lg.add_node(node)
    for i in range(4):
        lg.connect(ids[i], combine_node, 0, i)
    save_and_compare(kc, lg, combine_node, "value_node.png")


def _invert_graph(kc):
    g = kc.NodeGraph.new()
    white = g.add_node(kc.Node.new(kc.NodeType.Value(1.0)))
    inp = g.add_node(kc.Node.new(kc.NodeType.InputGray("in")))
    sub = g.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
    out = g.add_node(kc.Node.new(kc.NodeType.OutputGray("out")))
    g.connect(white, sub, 0, 0)
    g.connect(inp, sub, 0, 1)
    g.connect(sub, out, 0, 0)
    return g


@pytest.mark.parametrize("imported", [False, True])
def test_invert_graph_node(kc, imported):  # :991-1071, :1110-1160
    invert_graph = kc.NodeGraph.from_path(os.path.join(INPUTS, "invert_graph.json")) if imported else _invert_graph(kc)
    in_slot = invert_graph.input_slot_id_with_name("in")
    out_slot = invert_graph.output_slot_id_with_name("out")
    _, lg = new_graph(kc)
    image_node = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_2)))
    graph_node = lg.add_node(kc.Node.new(kc.NodeType.Graph(invert_graph)))
    separate_node = lg.add_node(kc.Node.new(kc.NodeType.SeparateRgba))
    output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputGray("out")))
    lg.connect(image_node, separate_node, 0, 0)
    lg.connect(separate_node, graph_node, 0, in_slot)
    lg.connect(graph_node, output_node, out_slot, 0)
    save_and_compare(kc, lg, output_node, "invert_graph_node_import.png" if imported else "invert_graph_node.png")


def test_temp_connect_while_updating(kc):  # :1162-1205
    _, lg = new_graph(kc)
    lg.auto_update = True
    lg.use_cache = True
    value_node = lg.add_node(kc.Node.new(kc.NodeType.Value(0.5)))
    combine_node = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
    separate_node = lg.add_node(kc.Node.new(kc.NodeType.SeparateRgba))
    lg.connect(combine_node, separate_node, 0, 0)
    lg.connect(value_node, combine_node, 0, 0)
    assert kc.LiveGraph.await_clean_read(lg, combine_node).slot_data_size(combine_node, 0) == (1, 1)


@pytest.mark.parametrize("rgba", [True, False])
def test_graph_node_passthrough(kc, rgba):  # :1207-1328
    nested = kc.NodeGraph.new()
    ni = nested.add_node(kc.Node.new(kc.NodeType.InputRgba("in") if rgba else kc.NodeType.InputGray("in")))
    no = nested.add_node(kc.Node.new(kc.NodeType.OutputRgba("out") if rgba else kc.NodeType.OutputGray("out")))
    nested.connect(ni, no, 0, 0)
    in_slot, out_slot = nested.input_slot_id_with_name("in"), nested.output_slot_id_with_name("out")
    _, lg = new_graph(kc)
    input_node = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_2)))
    if rgba:
        graph_node = lg.add_node(kc.Node.new(kc.NodeType.Graph(nested)))
        output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
        lg.connect(input_node, graph_node, 0, in_slot)
    else:
        separate_node = lg.add_node(kc.Node.new(kc.NodeType.SeparateRgba))
        graph_node = lg.add_node(kc.Node.new(kc.NodeType.Graph(nested)))
        output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputGray("out")))
        lg.connect(input_node, separate_node, 0, 0)
        lg.connect(separate_node, graph_node, 0, in_slot)
    lg.connect(graph_node, output_node, out_slot, 0)
    save_and_compare(kc, lg, output_node, "graph_node_rgba.png" if rgba else "graph_node_gray.png")


def test_height_to_normal_node(kc):  # :1349-1384
    _, lg = new_graph(kc)
    input_node = lg.add_node(kc.Node.new(kc.NodeType.Image(CLOUDS)))
    separate_node = lg.add_node(kc.Node.new(kc.NodeType.SeparateRgba))
    h2n_node = lg.add_node(kc.Node.new(kc.NodeType.HeightToNormal))
    output_node = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
    lg.connect(input_node, separate_node, 0, 0)
    lg.connect(separate_node, h2n_node, 0, 0)
    lg.connect(h2n_node, output_node, 0, 0)
    save_and_compare(kc, lg, output_node, "height_to_normal_node.png")


def test_read_dirty_read(kc):  # :1386-1437
    _, lg = new_graph(kc)
    lg.use_cache = True
    val_node = lg.add_node(kc.Node.new(kc.NodeType.Value(0.5)))
    combine_node = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
    lg.connect(val_node, combine_node, 0, 0)

    def verify_pixel(identifier):
        px = kc.LiveGraph.await_clean_read(lg, combine_node).slot_data(combine_node, 0).image.to_u8()
        assert px.reshape(-1).tolist() == [127, 0, 0, 255], identifier

    verify_pixel("Before dirty")
    lg.disconnect_slot(val_node, kc.Side.Output, 0)
    lg.connect(val_node, combine_node, 0, 0)
    verify_pixel("After dirty")


@pytest.mark.parametrize("mix", ["Add", "Subtract", "Multiply", "Divide", "Pow"])
@pytest.mark.parametrize("rgba", [False, True])
def test_mix_node(kc, mix, rgba):  # :1439-1568
    _, lg = new_graph(kc)
    if rgba:
        a = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_1)))
        b = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_2)))
        m = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.parse(mix))))
        out = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
        lg.connect(a, m, 0, 0)
        lg.connect(b, m, 0, 1)
    else:
        img = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_2)))
        sep = lg.add_node(kc.Node.new(kc.NodeType.SeparateRgba))
        m = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.parse(mix))))
        out = lg.add_node(kc.Node.new(kc.NodeType.OutputGray("out")))
        lg.connect(img, sep, 0, 0)
        lg.connect(sep, m, 0, 0)
        lg.connect(sep, m, 1, 1)
    lg.connect(m, out, 0, 0)
    name = "%s_node_%s.png" % ({"Add": "add", "Subtract": "subtract", "Multiply": "multiply", "Divide": "divide",
                                "Pow": "pow"}[mix], "rgba" if rgba else "gray")
    buf = kc.LiveGraph.await_clean_read(lg, out).buffer_rgba(out, 0)
    want = read_png(os.path.join(COMPARE, name))
    if mix == "Pow":
        assert np.abs(buf.astype(int) - want.astype(int)).max() <= 1 and (buf != want).sum() <= 8
    else:
        assert np.array_equal(buf, want)


def test_missing_image_is_magenta_pixel(kc):  # src/node/image.rs:13-18
    _, lg = new_graph(kc)
    img = lg.add_node(kc.Node.new(kc.NodeType.Image("does_not_exist.png")))
    out = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("out")))
    lg.connect(img, out, 0, 0)
    assert kc.LiveGraph.await_clean_read(lg, out).buffer_rgba(out, 0).reshape(-1).tolist() == [255, 0, 255, 255]


def test_write_node_roundtrip(kc, tmp_path):  # src/node/write.rs:5-21
    _, lg = new_graph(kc)
    img = lg.add_node(kc.Node.new(kc.NodeType.Image(IMAGE_2)))
    wr = lg.add_node(kc.Node.new(kc.NodeType.Write(str(tmp_path / "w.png"))))
    lg.connect(img, wr, 0, 0)
    kc.LiveGraph.await_clean_read(lg, wr)
    assert np.array_equal(read_png(str(tmp_path / "w.png")), read_png(os.path.join(INPUTS, IMAGE_2)))
