"""Host graph logic under random operation sequences, no GPU needed: a NodeGraph is driven through the C ABI
(add / remove nodes, connect / disconnect, nested Graph nodes, every node type, names that collide) while a
plain-Python model applies the reference's rules (src/node_graph.rs:315-446, src/node/mod.rs:209-221) to the
same sequence; after every operation both hold the same nodes and edges, and the serde JSON survives a round
trip byte for byte.  Run under ASan/UBSan by tools/sanitize_host.sh."""
import json

import numpy as np
import pytest

from kanter_core_amd import MixType, Node, NodeGraph, NodeType, ResizeFilter, ResizePolicy, Side, Size, SlotId, TexProError

G, R, X = "G", "R", "X"
IN_SLOTS = {"InputGray": "", "InputRgba": "", "OutputGray": G, "OutputRgba": R, "Image": "", "Embed": "", "Value": "",
            "Mix": X + X, "HeightToNormal": G, "SeparateRgba": R, "CombineRgba": G * 4}
OUT_SLOTS = {"InputGray": G, "InputRgba": R, "OutputGray": "", "OutputRgba": "", "Image": R, "Embed": R, "Value": G,
             "Mix": X, "HeightToNormal": R, "SeparateRgba": G * 4, "CombineRgba": R}


def fits(out_t, in_t):  # SlotType::fits, src/node/mod.rs:209-221
    return out_t == X or in_t == X or out_t == in_t


def make_node(rng, kind, counter):
    if kind in ("InputGray", "InputRgba", "OutputGray", "OutputRgba"):
        return Node.new(getattr(NodeType, kind)("io%d" % rng.integers(3)))  # colliding names on purpose
    if kind == "Image":
        return Node.new(NodeType.Image("img_%d.png" % counter))
    if kind == "Embed":
        return Node.new(NodeType.Embed(int(rng.integers(4))))
    if kind == "Value":
        return Node.new(NodeType.Value(float(np.float32(rng.random()))))
    if kind == "Mix":
        return Node.new(NodeType.Mix(int(rng.integers(5))))
    return Node.new(getattr(NodeType, kind))


@pytest.mark.parametrize("seed", range(40))
def test_random_operation_sequences_match_the_model(seed):
    rng = np.random.default_rng(0xF0250000 + seed)
    g = NodeGraph.new()
    kinds = {}   # node id -> kind
    edges = []   # (out, in, out_slot, in_slot) in insertion order
    counter = 0
    for _ in range(int(rng.integers(20, 80))):
        op = rng.random()
        if op < 0.35 or len(kinds) < 3:
            kind = list(IN_SLOTS)[rng.integers(len(IN_SLOTS))]
            node = make_node(rng, kind, counter).with_resize_policy(
                [ResizePolicy.MostPixels, ResizePolicy.SpecificSlot(SlotId(1)), ResizePolicy.SpecificSize(Size(7, 9))][rng.integers(3)]
            ).with_resize_filter(int(rng.integers(5)))
            counter += 1
            kinds[int(g.add_node(node))] = kind
        elif op < 0.75:
            ids = sorted(kinds)
            a, b = int(ids[rng.integers(len(ids))]), int(ids[rng.integers(len(ids))])
            so, si = int(rng.integers(5)), int(rng.integers(5))
            ok = so < len(OUT_SLOTS[kinds[a]]) and si < len(IN_SLOTS[kinds[b]]) and fits(OUT_SLOTS[kinds[a]][so], IN_SLOTS[kinds[b]][si])
            dup = (a, b, so, si) in edges
            try:
                g.connect(a, b, so, si)
                assert ok and not dup, ("connect accepted", kinds[a], so, kinds[b], si)
                edges[:] = [e for e in edges if not (e[1] == b and e[3] == si)]  # the input slot is vacated first (:416-446)
                edges.append((a, b, so, si))
            except TexProError:
                if ok and dup:  # the occupied-slot disconnect happens before the duplicate check
                    edges[:] = [e for e in edges if not (e[1] == b and e[3] == si)]
                else:
                    assert not ok, ("connect refused", kinds[a], so, kinds[b], si)
        elif op < 0.87:
            ids = sorted(kinds)
            n = int(ids[rng.integers(len(ids))])
            g.remove_node(n)
            del kinds[n]
            edges[:] = [e for e in edges if e[0] != n and e[1] != n]
        else:
            ids = sorted(kinds)
            n = int(ids[rng.integers(len(ids))])
            side, slot = (Side.Input, int(rng.integers(4))) if rng.random() < 0.5 else (Side.Output, int(rng.integers(4)))
            hit = [e for e in edges if (e[1] == n and e[3] == slot) if side == Side.Input] if side == Side.Input else \
                  [e for e in edges if e[0] == n and e[2] == slot]
            try:
                g.disconnect_slot(n, side, slot)
                assert hit
                edges[:] = [e for e in edges if e not in hit]
            except TexProError:
                assert not hit
        # the library's view equals the model's after every operation
        assert sorted(int(i) for i in g.node_ids()) == sorted(kinds)
        assert [(int(e.output_id), int(e.input_id), int(e.output_slot), int(e.input_slot)) for e in g.edges()] == edges
    text = g.to_json()
    again = NodeGraph.from_json(text)
    assert again.to_json() == text
    doc = json.loads(text)
    assert len(doc["nodes"]) == len(kinds) and len(doc["edges"]) == len(edges)
