"""GPU parity, differential: seeded random node graphs evaluated by the device evaluator (lazy Mix
chains, deferred resizes, fused launches) and by the oracle's literal process_node restatement
(oracle.RefGraph), every requested slot compared bit for bit.

The graphs mix everything the hot path has -- the five blend ops minus Pow (its <= 1 ulp tolerance
would be amplified by later nodes), invert, Value, SeparateRgba / CombineRgba aliasing, type changes,
HeightToNormal, all six resize policies and five filters, sources of different sizes and types --
wired at random, so they reach combinations the hand-written cases do not: fan-out from the middle of a
chain, the same plane on both sides, resized operands shared by several consumers, 1x1 constants
meeting images, chains cut by non-Mix consumers."""
import json

import numpy as np
import pytest

from util import assert_planes, bit_equal

pytestmark = pytest.mark.gpu

SIZES = [(24, 16), (48, 32), (17, 29), (5, 3), (1, 1), (64, 8), (300, 40), (1100, 9), (7, 260)]
OPS = ["Add", "Subtract", "Multiply", "Divide"]
FILTERS = ["Nearest", "Triangle", "CatmullRom", "Gaussian", "Lanczos3"]


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    return kc


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    return orc


def _source(rng, rgba):
    w, h = SIZES[rng.integers(len(SIZES))]
    planes = []
    for _ in range(4 if rgba else 1):
        p = (rng.random((h, w), dtype=np.float32) * np.float32(1.6) - np.float32(0.3)).astype(np.float32)
        if p.size >= 12 and rng.random() < 0.3:
            p.reshape(-1)[rng.integers(p.size, size=3)] = rng.choice(
                np.array([np.nan, np.inf, -np.inf, -0.0, 0.0, 1e-42, 3e38], np.float32), 3)
        planes.append(p)
    return planes


def _policy(kc, rng, n_inputs):
    k = rng.integers(6)
    P = kc.ResizePolicy
    if k == 4:
        return P.SpecificSlot(kc.SlotId(int(rng.integers(max(n_inputs, 1)))))
    if k == 5:
        w, h = SIZES[rng.integers(len(SIZES))]
        return P.SpecificSize(kc.Size(w + int(rng.integers(3)), h))
    return [P.MostPixels, P.LeastPixels, P.LargestAxes, P.SmallestAxes][k]


def _build(kc, orc, seed, info=None):
    """-> (live graph, RefGraph, node ids to request); `info` (optional dict) receives what the mutation
    fuzzer needs: embedded images, every output slot with its creator's position, every node with its slots."""
    rng = np.random.default_rng(seed)
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    embedded = {}
    # (node id, slot id, static slot type) of every output slot so far.  connect() checks the STATIC types
    # (src/node/mod.rs:209-221): "G" gray, "R" rgba, "X" gray-or-rgba (Mix).  What flows at run time may
    # differ (a gray image embedded behind Embed's rgba slot, an rgba Mix result into a gray slot): those
    # are the reference's run-time paths (four 1x1 zeros, empty result -> InvalidBufferCount) and stay in.
    outs = []
    for eid in range(int(rng.integers(2, 5))):
        rgba = bool(rng.random() < 0.6)
        planes = _source(rng, rgba)
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(planes)), eid)
        embedded[eid] = orc.Image([p.copy() for p in planes])
        n = lg.add_node(kc.Node.new(kc.NodeType.Embed(eid)))
        outs.append((n, 0, "R"))
    for _ in range(int(rng.integers(0, 3))):
        n = lg.add_node(kc.Node.new(kc.NodeType.Value(float(np.float32(rng.random() * 1.5 - 0.25)))))
        outs.append((n, 0, "G"))
    made = []

    def pick(slot_type):
        cands = [o for o in outs if slot_type == "X" or o[2] in (slot_type, "X")]
        return cands[rng.integers(len(cands))] if cands else None

    for _ in range(int(rng.integers(6, 30))):
        r = rng.random()
        if r < 0.62:
            node = kc.Node.new(kc.NodeType.Mix(kc.MixType.parse(OPS[rng.integers(len(OPS))])))
            ins = [pick("X") if rng.random() < 0.93 else None for _ in range(2)]
            produces = [(0, "X")]
        elif r < 0.72:
            node = kc.Node.new(kc.NodeType.SeparateRgba)
            ins = [pick("R")]
            produces = [(s_, "G") for s_ in range(4)]
        elif r < 0.84:
            node = kc.Node.new(kc.NodeType.CombineRgba)
            ins = [pick("G") if rng.random() < 0.8 else None for _ in range(4)]
            produces = [(0, "R")]
        elif r < 0.92:
            node = kc.Node.new(kc.NodeType.HeightToNormal)
            ins = [pick("G")]
            produces = [(0, "R")]
        else:  # invert: Mix(Subtract)(Value 1, x)
            one = lg.add_node(kc.Node.new(kc.NodeType.Value(1.0)))
            node = kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract))
            ins = [(one, 0, "G"), pick("X")]
            produces = [(0, "X")]
        n_in = sum(i is not None for i in ins)
        node = node.with_resize_policy(_policy(kc, rng, n_in)).with_resize_filter(
            kc.ResizeFilter.parse(FILTERS[rng.integers(len(FILTERS))]))
        n = lg.add_node(node)
        order = list(range(len(ins)))
        rng.shuffle(order)  # edge insertion order != slot order: calculate_size iterates edges
        for slot in order:
            if ins[slot] is not None:
                lg.connect(ins[slot][0], n, ins[slot][1], slot)
        outs.extend((n, s_, t) for s_, t in produces)
        made.append(n)
        if info is not None:
            slot_types = {"Mix": "XX", "SeparateRgba": "R", "CombineRgba": "GGGG", "HeightToNormal": "G"}
            kind = kc.NodeType._KINDS[node.node_type.kind]
            info.setdefault("nodes", []).append((n, kind, slot_types[kind], len(outs) - len(produces)))
    if info is not None:
        info["embedded"], info["outs"] = embedded, outs
    graph = json.loads(lg.node_graph().to_json())
    ref = orc.RefGraph(graph, embedded=embedded)
    k = min(len(made), int(rng.integers(1, 4)))
    req = [made[-1]] + [made[i] for i in rng.choice(len(made), size=k, replace=False)] if made else []
    return lg, ref, req


@pytest.mark.parametrize("seed", range(300))
def test_random_graph_matches_the_oracle(kc, orc, seed):
    _, _, requested = _build(kc, orc, 0xF0220000 + seed)
    for n in requested:
        # a fresh evaluation per requested node: with use_cache == false (the default) a node's planes are
        # dropped once all its children are processed (src/engine.rs:58-75), so a second request on the
        # same live graph may legitimately find a Clean node without data
        lg, ref, _ = _build(kc, orc, 0xF0220000 + seed)
        try:
            want = ref.node_slot_datas(int(n))
        except (RuntimeError, AssertionError) as e:
            # the reference fails this node: mixed types -> InvalidBufferCount (src/node/node_type.rs:134),
            # or panics on an rgba image in a CombineRgba slot (src/node/combine_rgba.rs:23)
            assert str(e) in ("InvalidBufferCount", "NodeProcessing") or "RGBA image into this slot" in str(e), e
            with pytest.raises(kc.TexProError):
                lg.await_clean(n)
            continue
        got = lg.await_clean(n).node_slot_datas(n)
        assert len(got) == len(want), (seed, int(n))
        for g, w in zip(sorted(got, key=lambda s: s.slot_id), sorted(want, key=lambda s: s.slot_id)):
            assert int(g.slot_id) == int(w.slot_id)
            assert g.image.is_rgba() == w.image.is_rgba, (seed, int(n))
            assert_planes(g.image.planes(), w.image.planes, what="seed %d node %d slot %d" % (seed, int(n), int(g.slot_id)))


@pytest.mark.parametrize("seed", [0xF0990000 + 11057])
def test_soak_seeds_that_found_defects(kc, orc, seed):
    """Seeds from profiles/soak_fuzz.py.  0xF0990000 + 11057: an image whose planes include a chain AND a plane that chain
    has to run first (more than KC_CHAIN_MAX_IN inputs): forcing the image forced the prefix through the recursion and then
    met it again, resident, in its own work list -- a null dereference in chain_flatten."""
    _, _, requested = _build(kc, orc, seed)
    for n in requested:
        lg, ref, _ = _build(kc, orc, seed)
        want = ref.node_slot_datas(int(n))
        got = lg.await_clean(n).node_slot_datas(n)
        assert len(got) == len(want)
        for g, w in zip(sorted(got, key=lambda s: s.slot_id), sorted(want, key=lambda s: s.slot_id)):
            assert_planes(g.image.planes(), w.image.planes, what="seed %x node %d" % (seed, int(n)))


@pytest.mark.parametrize("seed", range(40))
def test_random_graph_unfused_and_cached_agree(kc, orc, seed):
    """The same graphs with fusion switched off and with use_cache: every node materialised."""
    results = []
    for mode in ("fused", "unfused", "use_cache"):
        kc.set_fusion(mode != "unfused")
        try:
            _, _, requested = _build(kc, orc, 0xF0230000 + seed)
            planes = []
            for n in requested:
                lg, _, _ = _build(kc, orc, 0xF0230000 + seed)
                lg.use_cache = mode == "use_cache"
                try:
                    planes.append([sd.image.planes() for sd in sorted(lg.await_clean(n).node_slot_datas(n), key=lambda s: s.slot_id)])
                except kc.TexProError:
                    planes.append("error")
            results.append(planes)
        finally:
            kc.set_fusion(True)
    for other in results[1:]:
        assert len(other) == len(results[0])
        for a, b in zip(results[0], other):
            if isinstance(a, str) or isinstance(b, str):
                assert a == b
                continue
            for pa, pb in zip(a, b):
                assert_planes(pa, pb, what="seed %d" % seed)


def _compare(kc, orc, lg, embedded, n, what):
    ref = orc.RefGraph(json.loads(lg.node_graph().to_json()), embedded=embedded)
    try:
        want = ref.node_slot_datas(int(n))
    except (RuntimeError, AssertionError):
        with pytest.raises(kc.TexProError):
            lg.await_clean(n)
        return False
    got = lg.await_clean(n).node_slot_datas(n)
    assert len(got) == len(want), what
    for g, w in zip(sorted(got, key=lambda s: s.slot_id), sorted(want, key=lambda s: s.slot_id)):
        assert g.image.is_rgba() == w.image.is_rgba, what
        assert_planes(g.image.planes(), w.image.planes, what=what)
    return True


@pytest.mark.parametrize("seed", range(80))
def test_random_graph_edits_re_evaluate_like_a_fresh_graph(kc, orc, seed):
    """The LiveGraph state machine: after every edit (another blend op, an input rewired to a different
    producer, an edge removed, use_cache toggled, an intermediate result materialised behind the evaluator's
    back) the requested node is recomputed and equals a fresh evaluation of the edited graph by the oracle."""
    rng = np.random.default_rng(0xF0240000 + seed)
    info = {}
    lg, _, requested = _build(kc, orc, 0xF0240000 + seed, info)
    if not requested:
        return
    target = requested[0]
    if not _compare(kc, orc, lg, info["embedded"], target, "seed %d initial" % seed):
        return
    nodes, outs = info["nodes"], info["outs"]
    for step in range(5):
        n, kind, slots, first_out = nodes[rng.integers(len(nodes))]
        edit = rng.integers(5)
        if edit == 0 and kind == "Mix":
            lg.set_mix_type(n, kc.MixType.parse(OPS[rng.integers(len(OPS))]))
        elif edit == 1:
            slot = int(rng.integers(len(slots)))
            cands = [o for o in outs[:first_out] if slots[slot] == "X" or o[2] in (slots[slot], "X")]  # earlier producers only: no cycles
            if cands:
                src = cands[rng.integers(len(cands))]
                lg.connect(src[0], n, src[1], slot)
        elif edit == 2:
            try:
                lg.disconnect_slot(n, kc.Side.Input, int(rng.integers(len(slots))))
            except kc.TexProError:  # SlotNotOccupied, as the reference reports it
                pass
        elif edit == 3:
            lg.use_cache = not lg.use_cache
        else:
            try:
                for sd in lg.node_slot_datas(n):  # whatever the node still holds: force it into HBM in place
                    sd.image.materialize()
            except kc.TexProError:
                pass
        if not _compare(kc, orc, lg, info["embedded"], target, "seed %d after edit %d (%d on node %d)" % (seed, step, edit, int(n))):
            return


@pytest.mark.parametrize("seed", [210, 5678])
def test_soak_edit_seeds_that_found_defects(kc, orc, seed):
    """A source whose data had been dropped (use_cache == false) was re-dirtied WITH its descendants in the middle of a
    walk that had already passed some of them: they stayed Dirty above Clean children, and a later set_mix_type on one of
    them ("already Dirty") did not propagate -- the requested node kept its old pixels."""
    test_random_graph_edits_re_evaluate_like_a_fresh_graph(kc, orc, seed)


@pytest.mark.parametrize("seed", range(200))
def test_random_resizes_match_the_oracle(kc, orc, seed):
    """Random source and target extents (1 .. ~700, log-uniform, independent per axis) and filters: tile edges,
    partial 4-column groups, windows from 1 tap to hundreds, register-tap / LDS-table / two-pass forms."""
    rng = np.random.default_rng(0xF0260000 + seed)
    dim = lambda: int(np.exp(rng.uniform(0.0, np.log(700.0))))  # noqa: E731
    sw, sh, dw, dh = max(1, dim()), max(1, dim()), max(1, dim()), max(1, dim())
    filt = FILTERS[rng.integers(len(FILTERS))]
    p = (rng.random((sh, sw), dtype=np.float32) * np.float32(1.5) - np.float32(0.25)).astype(np.float32)
    if p.size >= 8:
        p.reshape(-1)[rng.integers(p.size, size=4)] = [np.nan, np.inf, -np.inf, -0.0]
    got = kc.resize_image(kc.SlotImage.from_planes([p]), (dw, dh), kc.ResizeFilter.parse(filt)).planes()[0]
    want = orc.resize_plane(p, dw, dh, filt)
    assert got.shape == want.shape
    # bit for bit; NaN payloads are not part of the contract (inf * 0 is -qNaN on x86, +qNaN on the GPU)
    assert bit_equal(got, want), "%s %dx%d -> %dx%d" % (filt, sw, sh, dw, dh)
