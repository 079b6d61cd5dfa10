"""A Mix whose two inputs are both chains that have not run keeps BOTH in one program (csrc/runtime.cpp plane_mix / join_ok,
chain_flatten; step code CH_SAVE_LOAD in csrc/chain_program.h) instead of running one of them on the spot: the reference's
per-node loops (src/node/mix.rs:136-192) applied to a fan-in without the plane in between.  The kernel compiled for the program
is the only one that can run it; while that kernel is not there (first sightings under the default mode, no hiprtc) the second
chain runs on its own exactly as before.  Everything bit for bit against the oracle and against join off."""
import numpy as np
import pytest

from test_gpu_replay import fanin_live, fanin_oracle
from util import SEED_A, SEED_B, assert_planes, splitmix_plane, synthetic_rgba

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    yield kc
    kc.set_option("join", 1)
    kc.set_specialize(1)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    return orc


def gray(seed, i, h, w):
    return splitmix_plane(seed, i, h, w) + np.float32(0.25)


@pytest.mark.parametrize("op", ["Add", "Subtract", "Multiply", "Divide", "Pow"])
@pytest.mark.parametrize("longer_left", [True, False])
def test_two_chains_meet_in_one_launch(kc, orc, op, longer_left):
    """x = (a * b) - c  (two steps),  y = d + e  (one step),  result = Mix(op)(x, y) or (y, x), then one more step on it."""
    h, w = 40, 72
    a, b, c, d, e, f = (gray(SEED_A, i, h, w) for i in range(6))
    mk = lambda p: kc.SlotImage.from_planes([p])  # noqa: E731
    M = lambda o, l, r: kc.mix_process(l, r, getattr(kc.MixType, o))  # noqa: E731
    kc.set_specialize(2)
    want_x = orc.mix_plane("Subtract", orc.mix_plane("Multiply", a, b), c)
    want_y = orc.mix_plane("Add", d, e)
    want = orc.mix_plane(op, want_x, want_y) if longer_left else orc.mix_plane(op, want_y, want_x)
    want = orc.mix_plane("Subtract", np.ones((h, w), np.float32), orc.mix_plane("Add", want, f))
    got = {}
    for join in (1, 0):
        kc.set_option("join", join)
        j0, l0 = kc.stats_counter("join_launches"), kc.stats()["kernel_launches"]
        x = M("Subtract", M("Multiply", mk(a), mk(b)), mk(c))
        y = M("Add", mk(d), mk(e))
        z = M(op, x, y) if longer_left else M(op, y, x)
        z = M("Subtract", kc.SlotImage.from_value((w, h), 1.0, False), M("Add", z, mk(f)))
        got[join] = z.planes()
        n_launch = kc.stats()["kernel_launches"] - l0
        # 6 inputs: more than a program holds (4), so the join takes place where it fits; what matters is fewer launches
        assert (kc.stats_counter("join_launches") - j0 > 0) == bool(join) or n_launch >= 2
    kc.set_option("join", 1)
    if op != "Pow":  # pow_positive is within 1 ulp of the oracle's libm and the steps after it stretch that: device paths only
        assert_planes(got[1], [want], what="join on, %s" % op)
    else:
        assert np.allclose(got[1][0], want, rtol=1e-5, atol=1e-6, equal_nan=True)
    assert_planes(got[0], got[1], what="join off == join on, %s" % op)


def test_four_inputs_two_chains_one_launch(kc, orc):
    h, w = 33, 50
    a, b, c, d = (gray(SEED_B, i, h, w) for i in range(4))
    mk = lambda p: kc.SlotImage.from_planes([p])  # noqa: E731
    M = lambda o, l, r: kc.mix_process(l, r, getattr(kc.MixType, o))  # noqa: E731
    kc.set_specialize(2)
    kc.set_option("join", 1)
    want = orc.mix_plane("Subtract", orc.mix_plane("Multiply", a, b), orc.mix_plane("Add", c, d))
    ia, ib, ic, id_ = mk(a), mk(b), mk(c), mk(d)
    ia.materialize(), ib.materialize(), ic.materialize(), id_.materialize()
    l0, j0 = kc.stats()["kernel_launches"], kc.stats_counter("join_launches")
    z = M("Subtract", M("Multiply", ia, ib), M("Add", ic, id_))
    got = z.planes()
    assert kc.stats()["kernel_launches"] - l0 == 1 and kc.stats_counter("join_launches") - j0 == 1
    assert_planes(got, [want], what="(a * b) - (c + d) in one launch")


def test_a_tree_over_four_chains_on_two_sources_is_one_launch(kc, orc):
    """((a*b - a) + (b*b + a)) / ((a - b)*a - (b + a)*b): seven chains, every Mix above the leaves joins two of them, the top one
    two chains that hold joins themselves -- two sources, so everything fits one program: one launch."""
    h, w = 37, 53
    a, b = gray(SEED_A, 40, h, w), gray(SEED_B, 41, h, w)
    mk = lambda p: kc.SlotImage.from_planes([p])  # noqa: E731
    M = lambda o, l, r: kc.mix_process(l, r, getattr(kc.MixType, o))  # noqa: E731
    O = orc.mix_plane
    want = O("Divide", O("Add", O("Subtract", O("Multiply", a, b), a), O("Add", O("Multiply", b, b), a)),
             O("Subtract", O("Multiply", O("Subtract", a, b), a), O("Multiply", O("Add", b, a), b)))
    kc.set_specialize(2)
    got = {}
    for join in (1, 0):
        kc.set_option("join", join)
        ia, ib = mk(a), mk(b)
        ia.materialize(), ib.materialize()
        l0 = kc.stats()["kernel_launches"]
        z = M("Divide", M("Add", M("Subtract", M("Multiply", ia, ib), ia), M("Add", M("Multiply", ib, ib), ia)),
              M("Subtract", M("Multiply", M("Subtract", ia, ib), ia), M("Multiply", M("Add", ib, ia), ib)))
        got[join] = z.planes()
        n = kc.stats()["kernel_launches"] - l0
        assert (n == 1) if join else (n == 4), (join, n)
    kc.set_option("join", 1)
    assert_planes(got[1], [want], what="tree of joins")
    assert_planes(got[0], got[1], what="join off == join on")


@pytest.mark.parametrize("mode,cache", [(2, True), (1, True), (1, False), (0, True)])
def test_config4_tree_with_and_without_its_kernels(kc, orc, mode, cache, tmp_path):
    """BASELINE config #4's shape: 8 branches + a Mix(Add) tree.  mode 2: every program compiled at first sight -- ONE launch
    instead of 9; mode 1 (the default): the first evaluations fall back (the second chain runs on its own), later ones use the
    kernels -- with no kernel cache anywhere the very first evaluation does not even try (plain chains), with one it does, in the
    hope of finding the programs' code objects there; mode 0: no joins are made."""
    h, w = 24, 72
    # (the specialiser's memory outlives a test: the mode-1 cases get programs nobody has compiled yet)
    sub_nodes = 16 if mode != 1 else (14 if cache else 12)
    if mode == 1:
        kc.specialize_wait()
        kc.kernel_cache_set_dir(str(tmp_path) if cache else "off")
        if cache:
            (tmp_path / ".populated").write_text("")  # a cache that holds something, though not these programs
    sources = [(synthetic_rgba(SEED_A + 10 + k, h, w), synthetic_rgba(SEED_B + 10 + k, h, w)) for k in range(8)]
    want = fanin_oracle(orc, sources, sub_nodes)
    kc.set_option("join", 1)
    kc.set_option("replay", 0)
    kc.set_specialize(mode)
    try:
        tp, lg, plugs, root = fanin_live(kc, sources, sub_nodes)
        launches, fallbacks = [], []
        for rep in range(5):
            for (na, first) in plugs:
                lg.connect(na, first, 0, 0)
            l0, f0 = kc.stats()["kernel_launches"], kc.stats_counter("join_fallbacks")
            got = lg.await_clean(root).slot_data(root, 0).image.planes()
            launches.append(kc.stats()["kernel_launches"] - l0)
            fallbacks.append(kc.stats_counter("join_fallbacks") - f0)
            assert_planes(got, want, what="mode %d, evaluation %d" % (mode, rep))
            if mode == 1:
                kc.specialize_wait()
        # with a kernel of its own the whole graph is ONE program: 16 planes, 78 records, joins inside joins inside joins
        if mode == 2:
            assert launches == [1] * 5 and fallbacks == [0] * 5, (launches, fallbacks)
        elif mode == 0:
            assert launches == [9] * 5 and fallbacks == [0] * 5, (launches, fallbacks)
        elif not cache:
            # no kernel cache: the first evaluation builds plain chains (a one-shot evaluation gains nothing from programs it cannot
            # compile in time), the second meets programs without kernels, later ones have them
            assert launches[0] == 9 and fallbacks[0] == 0 and fallbacks[1] > 0 and launches[-1] == 1 and fallbacks[-1] == 0, (launches, fallbacks)
        else:
            # a kernel cache that might hold the programs: the first evaluation builds them, finds nothing and falls back
            assert fallbacks[0] > 0 and launches[0] >= 9 and launches[-1] == 1 and fallbacks[-1] == 0, (launches, fallbacks)
    finally:
        kc.set_option("replay", 1)
        kc.set_specialize(1)
        if mode == 1:
            kc.specialize_wait()
            kc.kernel_cache_set_dir(None)


def test_join_then_replay(kc, orc):
    h, w = 16, 40
    sources = [(synthetic_rgba(SEED_A + 30 + k, h, w), synthetic_rgba(SEED_B + 30 + k, h, w)) for k in range(4)]
    want = fanin_oracle(orc, sources, 6)
    kc.set_option("join", 1)
    kc.set_option("replay", 1)
    kc.set_specialize(2)
    try:
        tp, lg, plugs, root = fanin_live(kc, sources, 6)
        r0 = kc.stats_counter("replayed_evaluations")
        for rep in range(6):
            for (na, first) in plugs:
                lg.connect(na, first, 0, 0)
            got = lg.await_clean(root).slot_data(root, 0).image.planes()
            assert_planes(got, want, what="evaluation %d" % rep)
        assert kc.stats_counter("replayed_evaluations") - r0 >= 3
        # the kernels go away under a recording that needs them: the next evaluation walks, and makes no joins
        kc.set_specialize(0)
        for rep in range(2):
            for (na, first) in plugs:
                lg.connect(na, first, 0, 0)
            got = lg.await_clean(root).slot_data(root, 0).image.planes()
            assert_planes(got, want, what="after kc_set_specialize(0), evaluation %d" % rep)
    finally:
        kc.set_specialize(1)


def _braided_graph(kc, seed):
    """A run of Mix nodes in which every node continues the previous result and takes ANY earlier result, source or constant as
    its other input (profiles/soak_replay.py's build_simple): chains meet their own prefixes and each other all the time."""
    rng = np.random.default_rng(seed)
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 70))
    outs = []
    for eid in range(int(rng.integers(1, 4))):
        planes = [(rng.random((h, w), dtype=np.float32) * np.float32(1.6) - np.float32(0.3)).astype(np.float32) for _ in range(4)]
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(planes)), eid)
        outs.append(lg.add_node(kc.Node.new(kc.NodeType.Embed(eid))))
    for _ in range(int(rng.integers(1, 3))):
        v = lg.add_node(kc.Node.new(kc.NodeType.Value(float(np.float32(rng.random())))))
        c = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
        for s_ in range(3):
            lg.connect(v, c, 0, s_)
        outs.append(c)
    prev = outs[0]
    ops = ["Add", "Subtract", "Multiply", "Divide"]
    plug = None
    for _ in range(int(rng.integers(2, 24))):
        n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.parse(ops[rng.integers(len(ops))]))))
        other = outs[rng.integers(len(outs))]
        slot = 0 if rng.random() < 0.5 else 1
        lg.connect(prev, n, 0, slot)
        lg.connect(other, n, 0, 1 - slot)
        plug = plug or (prev, n, slot)  # the first Mix's cable from the first source: re-plugging it dirties everything
        prev = n
        outs.append(n)
    return tp, lg, prev, plug


@pytest.mark.parametrize("seed", [1588953026] + list(range(7100, 7160)))
def test_braided_graphs_join_fallback_and_plain_agree(kc, seed):
    """Three ways to the same planes: joins with their kernels (mode 2), joins whose kernels are not there yet (mode 1, first
    sighting: the joined chains run on their own) and no joins.  Seed 1588953026 (profiles/soak_replay.py): a joined chain
    brought four inputs along, the chain it joined had been run meanwhile and counted as a fifth -- "cannot be split"."""
    results = []
    try:
        for mode, join in ((1, 1), (2, 1), (2, 0)):
            kc.set_specialize(mode)
            kc.set_option("join", join)
            kc.set_option("replay", 0)
            tp, lg, last, plug = _braided_graph(kc, seed)
            got = lg.await_clean(last).slot_data(last, 0).image.planes()
            if mode == 1:  # a graph's first evaluation builds plain chains: the second one meets the programs without kernels
                lg.connect(plug[0], plug[1], 0, plug[2])
                again = lg.await_clean(last).slot_data(last, 0).image.planes()
                assert_planes(again, got, what="second evaluation (fallback) vs first (plain chains), seed %d" % seed)
                got = again
            results.append(got)
    finally:
        kc.set_specialize(1)
        kc.set_option("join", 1)
        kc.set_option("replay", 1)
    assert_planes(results[0], results[2], what="first sighting (fallback) vs no joins, seed %d" % seed)
    assert_planes(results[1], results[2], what="joined kernels vs no joins, seed %d" % seed)
