"""Replay of a recorded evaluation (csrc/replay.cpp): an await_clean that repeats the previous one of the same node exactly
skips the node-by-node walk (the reference's src/engine.rs:200-307) and re-issues the recorded launches.  What must hold:
  * replayed results are the oracle's, bit for bit, every time;
  * everything observable afterwards (node states, which nodes hold slot data, the changed set) is what the walk leaves;
  * any edit -- a Mix type, an edge, an embedded image, a value, use_cache, a new node -- is seen: the next evaluation walks."""
import numpy as np
import pytest

from util import SEED_A, SEED_B, assert_planes, bit_equal, splitmix_plane, synthetic_rgba

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    yield kc
    kc.set_option("replay", 1)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    return orc


def chain_graph(kc, a, b, n):
    """The BASELINE chain (bench.py add_chain) on embedded images a, b."""
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 0)
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(b)), 1)
    na = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
    nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
    one = lg.add_node(kc.Node.new(kc.NodeType.Value(1.0)))
    white = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
    for s in range(3):
        lg.connect(one, white, 0, s)
    prev, first, mids = na, None, []
    for i in range(1, n + 1):
        if i & 1:
            x = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply if (i >> 1) & 1 else kc.MixType.Add)))
            lg.connect(prev, x, 0, 0)
            lg.connect(nb, x, 0, 1)
        else:
            x = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
            lg.connect(white, x, 0, 0)
            lg.connect(prev, x, 0, 1)
        first = first if first is not None else x
        mids.append(x)
        prev = x
    return tp, lg, na, nb, first, prev, mids


def snapshot(lg):
    ids = sorted(lg.node_ids())
    return ([(i, lg.node_state(i)) for i in ids], [(i, len(lg.node_slot_datas(i))) for i in ids], sorted(lg.changed_consume()))


@pytest.mark.parametrize("shape", [(64, 64), (40, 130)])
def test_replayed_evaluations_equal_the_walk_and_the_oracle(kc, orc, shape):
    h, w = shape
    a, b = synthetic_rgba(SEED_A, h, w), synthetic_rgba(SEED_B, h, w)
    want = orc.chain32(a, b, 12)
    kc.set_option("replay", 1)
    tp1, lg1, na1, _, first1, last1, _ = chain_graph(kc, a, b, 12)
    kc.set_option("replay", 0)
    tp2, lg2, na2, _, first2, last2, _ = chain_graph(kc, a, b, 12)
    n0 = kc.stats_counter("replayed_evaluations")
    for rep in range(6):
        kc.set_option("replay", 1)
        lg1.connect(na1, first1, 0, 0)
        got1 = lg1.await_clean(last1).slot_data(last1, 0).image.planes()
        s1 = snapshot(lg1)
        kc.set_option("replay", 0)
        lg2.connect(na2, first2, 0, 0)
        got2 = lg2.await_clean(last2).slot_data(last2, 0).image.planes()
        s2 = snapshot(lg2)
        assert_planes(got1, want, what="replay on, evaluation %d" % rep)
        assert_planes(got2, want, what="replay off, evaluation %d" % rep)
        assert s1 == s2, "evaluation %d: states / slot data / changed set differ between replay and walk" % rep
    kc.set_option("replay", 1)
    # the first evaluation builds everything, the second sees the steady-state starting point for the first time and is
    # recorded, from the third on every one is a replay
    assert kc.stats_counter("replayed_evaluations") - n0 >= 3


def test_every_kind_of_edit_is_seen(kc, orc):
    h, w = 48, 96
    a, b, c = synthetic_rgba(SEED_A, h, w), synthetic_rgba(SEED_B, h, w), synthetic_rgba(SEED_A + 77, h, w)
    kc.set_option("replay", 1)
    tp, lg, na, nb, first, last, mids = chain_graph(kc, a, b, 6)

    def run():
        lg.connect(na, first, 0, 0)
        return lg.await_clean(last).slot_data(last, 0).image.planes()

    def settle():
        n0 = kc.stats_counter("replayed_evaluations")
        for _ in range(4):
            got = run()
        assert kc.stats_counter("replayed_evaluations") > n0, "the steady state should be replaying"
        return got

    def ref(a_, b_, ops, b_first=None):
        x = a_[:3]
        one = np.ones((h, w), np.float32)
        for i, op in enumerate(ops, 1):
            bb = b_first if (i == 1 and b_first is not None) else b_
            x = [orc.mix_plane(op, x[ch], bb[ch]) for ch in range(3)] if i & 1 else [orc.mix_plane("Subtract", one, x[ch]) for ch in range(3)]
        return x + [one]

    ops = ["Add", "Subtract", "Multiply", "Subtract", "Add", "Subtract"]
    assert_planes(settle(), ref(a, b, ops), what="baseline")
    # 1. a Mix type changes
    lg.set_mix_type(mids[2], kc.MixType.Divide)
    ops[2] = "Divide"
    assert_planes(run(), ref(a, b, ops), what="after set_mix_type")
    assert_planes(settle(), ref(a, b, ops), what="after set_mix_type, replaying again")
    # 2. another source: a third embedded image feeds the first Mix instead of B
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(c)), 2)
    nc = lg.add_node(kc.Node.new(kc.NodeType.Embed(2)))
    lg.connect(nc, first, 0, 1)
    assert_planes(run(), ref(a, b, ops, b_first=c), what="after plugging another source")
    assert_planes(settle(), ref(a, b, ops, b_first=c), what="another source, replaying again")
    # 3. an edge moves: the first Mix reads C on both sides
    lg.connect(nc, first, 0, 0)
    got = lg.await_clean(last).slot_data(last, 0).image.planes()
    assert_planes(got, ref(c, b, ops, b_first=c), what="after moving an edge")
    lg.connect(na, first, 0, 0)
    assert_planes(settle(), ref(a, b, ops, b_first=c), what="edge moved back")
    # 4. use_cache on and off again
    lg.use_cache = True
    assert_planes(run(), ref(a, b, ops, b_first=c), what="use_cache on")
    lg.use_cache = False
    assert_planes(settle(), ref(a, b, ops, b_first=c), what="use_cache off again")
    # 5. a node is appended and requested instead
    tail = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
    lg.connect(last, tail, 0, 0)
    lg.connect(nb, tail, 0, 1)
    got = lg.await_clean(tail).slot_data(tail, 0).image.planes()
    want = ref(a, b, ops, b_first=c)
    assert_planes(got, [orc.mix_plane("Add", want[ch], b[ch]) for ch in range(3)] + [np.ones((h, w), np.float32)], what="appended node")
    # ... and the old request still works (the new node stays Dirty, the recording of `last` no longer matches: walk)
    assert_planes(run(), want, what="old node after the append")
    # 6. a Value changes: the invert constant becomes 0.5 (clamped by the implicit resize to [0, 1], still 0.5)
    # (Value nodes are parameters of the graph: a new node with another value replaces the old one's edges)
    half = lg.add_node(kc.Node.new(kc.NodeType.Value(0.5)))
    white2 = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
    for s_ in range(3):
        lg.connect(half, white2, 0, s_)
    lg.connect(white2, mids[1], 0, 0)
    got = run()
    x = a[:3]
    for i, op in enumerate(ops, 1):
        bb = c if i == 1 else b
        cst = np.full((h, w), 0.5 if i == 2 else 1.0, np.float32)
        x = [orc.mix_plane(op, x[ch], bb[ch]) for ch in range(3)] if i & 1 else [orc.mix_plane("Subtract", cst, x[ch]) for ch in range(3)]
    assert_planes(got, x + [np.ones((h, w), np.float32)], what="after changing a constant")


def test_replay_off_switch_and_counter(kc):
    h, w = 16, 32
    a, b = synthetic_rgba(SEED_A, h, w), synthetic_rgba(SEED_B, h, w)
    kc.set_option("replay", 0)
    try:
        tp, lg, na, nb, first, last, _ = chain_graph(kc, a, b, 4)
        n0 = kc.stats_counter("replayed_evaluations")
        for _ in range(5):
            lg.connect(na, first, 0, 0)
            lg.await_clean(last)
        assert kc.stats_counter("replayed_evaluations") == n0
        assert kc.get_option("replay") == 0
    finally:
        kc.set_option("replay", 1)


def fanin_live(kc, sources, sub_nodes):
    """Config #4's shape: one BASELINE chain per (a, b) pair of `sources`, joined by a fixed-order Mix(Add) tree."""
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    plugs, lasts = [], []
    for k, (a, b) in enumerate(sources):
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 2 * k)
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(b)), 2 * k + 1)
        na = lg.add_node(kc.Node.new(kc.NodeType.Embed(2 * k)))
        nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(2 * k + 1)))
        one = lg.add_node(kc.Node.new(kc.NodeType.Value(1.0)))
        white = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
        for s in range(3):
            lg.connect(one, white, 0, s)
        prev, first = na, None
        for i in range(1, sub_nodes + 1):
            if i & 1:
                x = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply if (i >> 1) & 1 else kc.MixType.Add)))
                lg.connect(prev, x, 0, 0)
                lg.connect(nb, x, 0, 1)
            else:
                x = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
                lg.connect(white, x, 0, 0)
                lg.connect(prev, x, 0, 1)
            first = first if first is not None else x
            prev = x
        plugs.append((na, first))
        lasts.append(prev)
    level = list(lasts)
    while len(level) > 1:
        nxt = []
        for i in range(0, len(level) - 1, 2):
            n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
            lg.connect(level[i], n, 0, 0)
            lg.connect(level[i + 1], n, 0, 1)
            nxt.append(n)
        if len(level) & 1:
            nxt.append(level[-1])
        level = nxt
    return tp, lg, plugs, level[0]


def fanin_oracle(orc, sources, sub_nodes):
    h, w = sources[0][0][0].shape
    level = [orc.chain32(a, b, sub_nodes)[:3] for (a, b) in sources]
    while len(level) > 1:
        nxt = [[orc.mix_plane("Add", level[i][c], level[i + 1][c]) for c in range(3)] for i in range(0, len(level) - 1, 2)]
        if len(level) & 1:
            nxt.append(level[-1])
        level = nxt
    return level[0] + [np.ones((h, w), np.float32)]


@pytest.mark.parametrize("n_branches,sub_nodes", [(8, 16), (5, 6), (2, 1)])
def test_an_evaluation_of_several_launches_is_replayed(kc, orc, n_branches, sub_nodes):
    """Config #4 on one device is nine launches (eight branches, then the add tree continuing from them): the recording
    holds all of them, and which input of a later launch is which result of an earlier one."""
    h, w = 24, 72
    sources = [(synthetic_rgba(SEED_A + k, h, w), synthetic_rgba(SEED_B + k, h, w)) for k in range(n_branches)]
    want = fanin_oracle(orc, sources, sub_nodes)
    # programs that join two chains (tests/test_gpu_join.py) are compiled at first sight here: while such a kernel is still
    # being compiled an evaluation falls back to separate launches and is deliberately not recorded
    kc.set_specialize(2)
    kc.set_option("wide", 0)  # programs of 4 input planes: the graph stays a SEQUENCE of launches, which is what this test replays
    kc.set_option("replay", 1)
    tp1, lg1, plugs1, root1 = fanin_live(kc, sources, sub_nodes)
    kc.set_option("replay", 0)
    tp2, lg2, plugs2, root2 = fanin_live(kc, sources, sub_nodes)
    n0 = kc.stats_counter("replayed_evaluations")
    launches = []
    for rep in range(6):
        kc.set_option("replay", 1)
        for (na, first) in plugs1:
            lg1.connect(na, first, 0, 0)
        l0 = kc.stats()["kernel_launches"]
        got1 = lg1.await_clean(root1).slot_data(root1, 0).image.planes()
        launches.append(kc.stats()["kernel_launches"] - l0)
        s1 = snapshot(lg1)
        kc.set_option("replay", 0)
        for (na, first) in plugs2:
            lg2.connect(na, first, 0, 0)
        got2 = lg2.await_clean(root2).slot_data(root2, 0).image.planes()
        s2 = snapshot(lg2)
        assert_planes(got1, want, what="replay on, evaluation %d" % rep)
        assert_planes(got2, want, what="replay off, evaluation %d" % rep)
        assert s1 == s2, "evaluation %d: states / slot data / changed set differ between replay and walk" % rep
    kc.set_option("replay", 1)
    assert kc.stats_counter("replayed_evaluations") - n0 >= 3
    assert len(set(launches[1:])) == 1 and launches[1] >= 1, launches  # a replay issues the launches the walk issued
    if n_branches > 2:  # (two short branches and their Mix are ONE launch once the program that joins them has its kernel)
        assert launches[1] > 1
    # only SOME branches are re-plugged: another starting point, so the walk runs (and is recorded in its turn)
    kc.set_option("replay", 1)
    for rep in range(3):
        lg1.connect(plugs1[0][0], plugs1[0][1], 0, 0)
        got = lg1.await_clean(root1).slot_data(root1, 0).image.planes()
        assert_planes(got, want, what="one branch re-plugged, evaluation %d" % rep)
    # a source is replaced: the recording holds the old image and must not be used
    c = synthetic_rgba(SEED_A + 991, h, w)
    lg1.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(c)), 2 * n_branches)
    nc = lg1.add_node(kc.Node.new(kc.NodeType.Embed(2 * n_branches)))
    plugs1[0] = (nc, plugs1[0][1])
    sources[0] = (c, sources[0][1])
    want = fanin_oracle(orc, sources, sub_nodes)
    for rep in range(4):
        for (na, first) in plugs1:
            lg1.connect(na, first, 0, 0)
        got = lg1.await_clean(root1).slot_data(root1, 0).image.planes()
        assert_planes(got, want, what="another source, evaluation %d" % rep)
    kc.set_specialize(1)
    kc.set_option("wide", 1)


def test_lazy_embedded_image_forced_from_outside_is_never_replayed_on_freed_operands(kc, orc):
    """An embedded image may be an unevaluated chain (mix_process result) whose operands only its own links keep alive.  The
    recorded launches of such an evaluation read those operands; once the image has been forced from outside (an export) the
    operands go back to the pool and may be recycled.  The recording must not be kept (or must keep the operands itself):
    every later evaluation equals the oracle."""
    h, w = 48, 96
    a, b = synthetic_rgba(SEED_A, h, w), synthetic_rgba(SEED_B, h, w)
    c = synthetic_rgba(0x5EED0003, h, w)
    kc.set_option("replay", 1)
    ia, ib = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b)
    lazy = kc.mix_process(ia, ib, kc.MixType.Add)  # not evaluated yet
    del ia, ib  # the chain's links are now the only owners of A's and B's planes
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, lazy), 0)
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(c)), 1)
    ne = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
    nc = lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
    m1 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply)))
    m2 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
    lg.connect(ne, m1, 0, 0)
    lg.connect(nc, m1, 0, 1)
    lg.connect(m1, m2, 0, 0)
    lg.connect(nc, m2, 0, 1)
    s = [orc.mix_plane("Add", a[k], b[k]) for k in range(3)]
    want = [orc.mix_plane("Subtract", orc.mix_plane("Multiply", s[k], c[k]), c[k]) for k in range(3)] + [np.ones((h, w), np.float32)]
    for rep in range(4):  # reach the steady state (replaying, if the recording qualified)
        lg.connect(ne, m1, 0, 0)
        assert_planes(lg.await_clean(m2).slot_data(m2, 0).image.planes(), want, what="evaluation %d" % rep)
    # force the embedded image from outside: its chain runs, its links (and with them A and B) are released
    assert_planes(lazy.planes(), s + [np.ones((h, w), np.float32)], what="the embedded image itself")
    kc.pool_trim()  # the operand blocks are hipFree'd ...
    junk = [kc.SlotImage.from_planes([np.full((h, w), np.nan, np.float32)] * 4) for _ in range(4)]  # ... and their addresses recycled
    for rep in range(3):
        lg.connect(ne, m1, 0, 0)
        assert_planes(lg.await_clean(m2).slot_data(m2, 0).image.planes(), want, what="after the outside force, evaluation %d" % rep)
    del junk
