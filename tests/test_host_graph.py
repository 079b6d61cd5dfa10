"""Host graph logic through the C ABI, no GPU needed: NodeGraph / LiveGraph structure, slot-type
rules, naming, JSON wire format, dirty-state propagation.  Mirrors the structural tests of the
reference (tests/integration_tests.rs: connect_invalid_slot :788-810, wrong_slot_type :1331-1347,
remove_node :772-785, unconnected :555-565, invert_graph_node_export :1073-1108)."""
import json
import os

import pytest

import kanter_core_amd as kc
from kanter_core_amd import (MixType, Node, NodeGraph, NodeState, NodeType, ResizeFilter, ResizePolicy, Side, Size,
                             SlotId, TexProError)
from golden_graphs import INPUTS


class HostOnlyTexPro:
    """TextureProcessor.new() binds a GPU; structural tests only need live graphs."""

    def __init__(self):
        import ctypes as C
        from kanter_core_amd import _lib
        self._h = C.c_void_p()
        assert _lib.load().kc_tex_pro_new(10_000_000, C.byref(self._h)) == 0

    def new_live_graph(self):
        import ctypes as C
        from kanter_core_amd import _lib
        h = C.c_void_p()
        assert _lib.load().kc_tex_pro_new_live_graph(self._h, C.byref(h)) == 0
        return kc.LiveGraph(h.value, self)


@pytest.fixture
def live_graph():
    return HostOnlyTexPro().new_live_graph()


def invert_graph():
    g = NodeGraph.new()
    white = g.add_node(Node.new(NodeType.Value(1.0)))
    inp = g.add_node(Node.new(NodeType.InputGray("in")))
    sub = g.add_node(Node.new(NodeType.Mix(MixType.Subtract)))
    out = g.add_node(Node.new(NodeType.OutputGray("out")))
    g.connect(white, sub, SlotId(0), SlotId(0))
    g.connect(inp, sub, SlotId(0), SlotId(1))
    g.connect(sub, out, SlotId(0), SlotId(0))
    return g


def test_invert_graph_export_matches_reference_file_byte_for_byte(tmp_path):
    # data/invert_graph.json was written by the reference's export_json; ids differ, shape must not
    ref = json.load(open(os.path.join(INPUTS, "invert_graph.json")))
    ours = json.loads(invert_graph().to_json())
    assert [n["node_type"] for n in ours["nodes"]] == [n["node_type"] for n in ref["nodes"]]
    assert [set(n) for n in ours["nodes"]] == [set(n) for n in ref["nodes"]]
    assert [(e["output_slot"], e["input_slot"]) for e in ours["edges"]] == [(e["output_slot"], e["input_slot"]) for e in ref["edges"]]
    loaded = NodeGraph.from_path(os.path.join(INPUTS, "invert_graph.json"))
    assert loaded.to_json() == open(os.path.join(INPUTS, "invert_graph.json")).read()
    p = tmp_path / "out.json"
    loaded.export_json(p)
    assert NodeGraph.from_path(p).to_json() == loaded.to_json()
    assert loaded.input_slot_id_with_name("in") == 808182335
    assert loaded.output_slot_id_with_name("out") == 3948812722
    # from_path continues numbering after the largest id (node_graph.rs:36-43)
    assert loaded.add_node(Node.new(NodeType.Value(0.0))) == 3948812723


def test_json_covers_every_node_type_and_policy():
    g = NodeGraph.new()
    inner = invert_graph()
    types = [NodeType.InputGray("a"), NodeType.InputRgba("b"), NodeType.OutputGray("c"), NodeType.OutputRgba("d"),
             NodeType.Graph(inner), NodeType.Image("x/y.png"), NodeType.Embed(7), NodeType.Write("o.png"),
             NodeType.Value(0.33), NodeType.Mix(MixType.Pow), NodeType.HeightToNormal, NodeType.SeparateRgba,
             NodeType.CombineRgba]
    for t in types:
        g.add_node(Node.new(t))
    n = Node.new(NodeType.Mix(MixType.Divide))
    n.resize_policy = ResizePolicy.SpecificSize(Size(256, 128))
    n.resize_filter = ResizeFilter.Lanczos3
    g.add_node(n)
    n = Node.new(NodeType.CombineRgba)
    n.resize_policy = ResizePolicy.SpecificSlot(SlotId(2))
    n.resize_filter = ResizeFilter.Nearest
    g.add_node(n)
    d = json.loads(g.to_json())
    tags = [t if isinstance(t, str) else list(t)[0] for t in (x["node_type"] for x in d["nodes"])]
    assert tags[:13] == list(NodeType._KINDS)
    assert d["nodes"][8]["node_type"] == {"Value": 0.33}
    assert d["nodes"][4]["node_type"]["Graph"]["nodes"][2]["node_type"] == {"Mix": "Subtract"}
    assert d["nodes"][13]["resize_policy"] == {"SpecificSize": {"width": 256, "height": 128}}
    assert d["nodes"][13]["resize_filter"] == "Lanczos3"
    assert d["nodes"][14]["resize_policy"] == {"SpecificSlot": 2}
    assert NodeGraph.from_json(g.to_json()).to_json() == g.to_json()
    with pytest.raises(TexProError):
        NodeGraph.from_json('{"nodes": [{"node_id": 0}], "edges": []}')
    with pytest.raises(TexProError):
        NodeGraph.from_json("not json")


def test_connect_invalid_slot(live_graph):
    value_node = live_graph.add_node(Node.new(NodeType.Value(0.0)))
    mix = live_graph.add_node(Node.new(NodeType.Mix(MixType.default())))
    live_graph.connect(value_node, mix, SlotId(0), SlotId(0))
    live_graph.connect(value_node, mix, SlotId(0), SlotId(1))
    with pytest.raises(TexProError) as e:
        live_graph.connect(value_node, mix, SlotId(0), SlotId(2))
    assert e.value.kind == "InvalidSlotId"


def test_wrong_slot_type(live_graph):
    image_node = live_graph.add_node(Node.new(NodeType.Image("image_1.png")))
    gray_node = live_graph.add_node(Node.new(NodeType.OutputGray("out")))
    with pytest.raises(TexProError) as e:  # the reference test unwraps this Err and panics
        live_graph.connect(image_node, gray_node, SlotId(0), SlotId(0))
    assert e.value.kind == "InvalidSlotType"


def test_remove_node_and_unconnected(live_graph):
    v = live_graph.add_node(Node.new(NodeType.Value(0.0)))
    live_graph.remove_node(v)
    assert live_graph.node_ids() == []
    live_graph.add_node(Node.new(NodeType.OutputRgba("out")))
    with pytest.raises(TexProError) as e:
        live_graph.remove_node(999)
    assert e.value.kind == "InvalidNodeId"


def test_node_ids_names_and_edge_replacement():
    g = NodeGraph.new()
    a = g.add_node(Node.new(NodeType.OutputGray("out")))
    b = g.add_node(Node.new(NodeType.OutputRgba("out")))
    c = g.add_node(Node.new(NodeType.OutputGray("out")))
    d = g.add_node(Node.new(NodeType.InputGray("")))
    assert (a, b, c, d) == (0, 1, 2, 3)
    names = [list(n["node_type"].values())[0] for n in json.loads(g.to_json())["nodes"]]
    assert names == ["out", "out_0", "out_1", "untitled"]  # avoid_name_collision, node_graph.rs:141-164
    with pytest.raises(TexProError):
        g.add_node_with_id(Node.with_id(NodeType.Value(1.0), 2))
    g.add_node_with_id(Node.with_id(NodeType.Value(1.0), 40))
    v2 = g.add_node(Node.new(NodeType.Value(2.0)))
    assert v2 == 4
    mix = g.add_node(Node.new(NodeType.Mix(MixType.Add)))
    g.connect(40, mix, 0, 0)
    g.connect(v2, mix, 0, 0)  # forces the connection: the old edge into the slot is dropped (:416-446)
    assert [(e.output_id, e.input_slot) for e in g.edges()] == [(v2, 0)]
    with pytest.raises(TexProError) as e:
        g.try_connect(40, mix, 0, 0)
    assert e.value.kind == "SlotOccupied"
    g.try_connect(40, mix, 0, 1)
    with pytest.raises(TexProError) as e:
        g.disconnect_slot(mix, Side.Output, 0)
    assert e.value.kind == "SlotNotOccupied"
    g.set_mix_type(mix, MixType.Pow)
    assert json.loads(g.to_json())["nodes"][-1]["node_type"] == {"Mix": "Pow"}
    with pytest.raises(TexProError):
        g.set_mix_type(v2, MixType.Pow)


def test_dirty_state_propagates_to_children(live_graph):
    v = live_graph.add_node(Node.new(NodeType.Value(0.5)))
    comb = live_graph.add_node(Node.new(NodeType.CombineRgba))
    sep = live_graph.add_node(Node.new(NodeType.SeparateRgba))
    out = live_graph.add_node(Node.new(NodeType.OutputGray("out")))
    live_graph.connect(v, comb, 0, 0)
    live_graph.connect(comb, sep, 0, 0)
    live_graph.connect(sep, out, 2, 0)
    assert all(live_graph.node_state(n) == NodeState.Dirty for n in (v, comb, sep, out))
    # Value / Combine / Separate / Output are pure plane aliasing: no kernel, so this evaluates
    # without a device -- and exercises the scheduler bookkeeping of src/engine.rs:34-103
    live_graph.await_clean(out)
    assert live_graph.node_state(out) == NodeState.Clean
    assert live_graph.slot_data_size(out, 0) == (1, 1)
    with pytest.raises(TexProError) as e:  # use_cache == false: parents were dropped (engine.rs:58-75)
        live_graph.slot_data(comb, 0)
    assert e.value.kind == "NoSlotData"
    assert sorted(live_graph.changed_consume()) == [v, comb, sep, out]
    live_graph.connect(v, comb, 0, 1)  # dirties comb and everything downstream
    assert live_graph.node_state(v) == NodeState.Clean
    assert [live_graph.node_state(n) for n in (comb, sep, out)] == [NodeState.Dirty] * 3
    live_graph.request(out)
    assert live_graph.node_state(out) == NodeState.Requested
    live_graph.prioritise(out)
    assert live_graph.node_state(out) == NodeState.Prioritised
    live_graph.update()
    assert live_graph.node_state(out) == NodeState.Clean


def test_use_cache_keeps_parents(live_graph):
    # use_cache / no_cache, tests/integration_tests.rs:249-305
    v = live_graph.add_node(Node.new(NodeType.Value(1.0)))
    out = live_graph.add_node(Node.new(NodeType.OutputGray("out")))
    live_graph.connect(v, out, 0, 0)
    live_graph.await_clean(out)
    with pytest.raises(TexProError):
        live_graph.slot_data(v, 0)
    live_graph.use_cache = True
    live_graph.connect(v, out, 0, 0)
    live_graph.await_clean(out)
    assert live_graph.slot_data(v, 0).size() == (1, 1)


def test_calculate_size_is_pure_host():
    P = ResizePolicy
    sizes = [(128, 128), (256, 256)]
    assert kc.calculate_size(P.LeastPixels, sizes) == (128, 128)
    assert kc.calculate_size(P.MostPixels, sizes) == (256, 256)
    assert kc.calculate_size(P.MostPixels, [(4, 4), (2, 8)]) == (2, 8)   # max_by keeps the last maximum
    assert kc.calculate_size(P.LeastPixels, [(4, 4), (2, 8)]) == (4, 4)  # min_by keeps the first minimum
    assert kc.calculate_size(P.LargestAxes, [(128, 64), (64, 128)]) == (128, 128)
    assert kc.calculate_size(P.SmallestAxes, [(128, 64), (64, 128)]) == (64, 64)
    assert kc.calculate_size(P.SpecificSlot(1), sizes, slot_index=1) == (256, 256)
    assert kc.calculate_size(P.SpecificSlot(1), sizes, slot_index=-1) == (1, 1)


def test_rename_output_and_image_path(live_graph):
    g = NodeGraph.new()
    a = g.add_node(Node.new(NodeType.OutputGray("out")))
    b = g.add_node(Node.new(NodeType.OutputRgba("out")))   # becomes out_0
    img = g.add_node(Node.new(NodeType.Image("a.png")))
    assert g.rename_output_node(b, "out") == "out_0"        # collides with a -> out_0 again
    assert g.rename_output_node(a, "albedo") == "out"
    assert g.rename_output_node(b, "out") == "out_0"        # "out" is free now
    names = [list(n["node_type"].values())[0] for n in json.loads(g.to_json())["nodes"]]
    assert names == ["albedo", "out", "a.png"]
    g.set_image_node_path(img, "b.png")
    assert json.loads(g.to_json())["nodes"][2]["node_type"] == {"Image": "b.png"}
    with pytest.raises(TexProError):
        g.rename_output_node(img, "x")
    with pytest.raises(TexProError):
        g.set_image_node_path(a, "x.png")
    o = live_graph.add_node(Node.new(NodeType.OutputGray("out")))
    assert live_graph.rename_output_node(o, "height") == "out"


def test_try_buffer_rgba_requests_when_not_clean(live_graph):
    v = live_graph.add_node(Node.new(NodeType.Value(1.0)))
    out = live_graph.add_node(Node.new(NodeType.OutputGray("out")))
    live_graph.connect(v, out, 0, 0)
    with pytest.raises(TexProError) as e:
        kc.LiveGraph.try_buffer_rgba(live_graph, out, 0)
    assert e.value.kind == "InvalidNodeId" and live_graph.node_state(out) == NodeState.Requested
    live_graph.update()
    assert live_graph.node_state(out) == NodeState.Clean


def _mix():
    return Node.new(NodeType.Mix(MixType.Add))


def test_cyclic_graph_is_an_error_not_a_crash(live_graph):
    # connect() accepts a cycle exactly as the reference's does (node_graph.rs:416-446 has no check); the
    # reference then recurses without bound in set_state / get_children_recursive.  Here every walk
    # terminates and asking for a node on the cycle reports it.
    lg = live_graph
    a, b, c = lg.add_node(_mix()), lg.add_node(_mix()), lg.add_node(_mix())
    lg.connect(a, b, SlotId(0), SlotId(0))
    lg.connect(b, c, SlotId(0), SlotId(0))
    lg.connect(c, a, SlotId(0), SlotId(0))
    for n in (a, b, c):
        assert lg.node_state(n) == NodeState.Dirty
    with pytest.raises(TexProError) as e:
        lg.await_clean(b)
    assert "cycle" in str(e.value)
    # the mutators that walk descendants terminate too
    lg.disconnect_slot(b, Side.Input, SlotId(0))
    lg.connect(a, b, SlotId(0), SlotId(0))
    lg.remove_node(c)
    assert lg.node_state(a) == NodeState.Dirty


def test_self_loop_is_reported(live_graph):
    lg = live_graph
    a = lg.add_node(_mix())
    lg.connect(a, a, SlotId(0), SlotId(1))
    with pytest.raises(TexProError):
        lg.await_clean(a)


def test_cycle_in_json_graph_is_reported(live_graph):
    g = NodeGraph.new()
    a, b = g.add_node(_mix()), g.add_node(_mix())
    g.connect(a, b, SlotId(0), SlotId(0))
    g.connect(b, a, SlotId(0), SlotId(0))
    again = NodeGraph.from_json(g.to_json())
    lg = live_graph
    lg.set_node_graph(again)
    with pytest.raises(TexProError):
        lg.await_clean(b)


def test_dirty_propagation_walks_a_very_long_chain_without_recursion(live_graph):
    # 20 000 nodes: also pins that building a graph is linear (appending a node or an edge patches the look-up index;
    # rebuilding it per connect() made this take two minutes)
    import time
    lg = live_graph
    n = 20000
    t0 = time.perf_counter()
    ids = [lg.add_node(_mix())]
    for _ in range(n):
        nxt = lg.add_node(_mix())
        lg.connect(ids[-1], nxt, SlotId(0), SlotId(0))
        ids.append(nxt)
    assert time.perf_counter() - t0 < 20.0
    lg.changed_consume()
    lg.connect(ids[0], ids[1], SlotId(0), SlotId(1))
    assert lg.node_state(ids[-1]) == NodeState.Dirty
