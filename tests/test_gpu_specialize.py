"""Run-time specialised chain kernels (csrc/specialize.cpp) against the step-table interpreter and the oracle:
the same programs evaluated with specialisation off (mode 0), compiled at first sight (mode 2) and compiled in
the background (mode 1) must give the same bits -- Add/Sub/Mul/Div exactly the oracle's, Pow within 1 ulp of it
and identical between the two device paths."""
import numpy as np
import pytest

from util import SEED_A, SEED_B, assert_planes, bit_equal, splitmix_plane, with_edge_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    old = kc.get_specialize()
    yield kc
    kc.set_specialize(old)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    return orc


def both_modes(kc, build):
    """build() -> SlotImage (lazy); returns (interpreter planes, specialised planes, specialised launches).
    One-step programs would run their ahead-of-time kernels (chain1.hip) before either: switched off here, see
    test_single_mix_ahead_of_time_kernels for that route."""
    kc.set_option("chain1", 0)
    try:
        kc.set_specialize(0)
        a = build().planes()
        kc.set_specialize(2)
        n0 = kc.specialize_stats()
        b = build().planes()
        n1 = kc.specialize_stats()
    finally:
        kc.set_option("chain1", 1)
    assert n1["compiles_failed"] == n0["compiles_failed"], "a specialised kernel failed to compile"
    return a, b, n1["specialized_launches"] - n0["specialized_launches"]


def chain32(kc, x, bb, n):
    w, h = x.size()
    white = kc.combine_rgba_process([kc.value_process(1.0)] * 3 + [None])
    for i in range(1, n + 1):
        if i & 1:
            x = kc.mix_process(x, bb, kc.MixType.Multiply if (i >> 1) & 1 else kc.MixType.Add)
        else:
            x = kc.mix_process(kc.resize_image(white, (w, h)), x, kc.MixType.Subtract)
    return x


@pytest.mark.parametrize("shape", [(256, 512), (97, 130), (64, 4100)])
def test_baseline_chain_specialised_equals_interpreter_equals_oracle(kc, orc, shape):
    h, w = shape  # 512: dense rows (flat program); 130 / 4100: pitched planes (row / column split)
    a = [with_edge_cases(splitmix_plane(SEED_A, c, h, w), c) for c in range(3)] + [np.ones((h, w), np.float32)]
    b = [with_edge_cases(splitmix_plane(SEED_B, c, h, w), c + 1) for c in range(3)] + [np.ones((h, w), np.float32)]
    ia, ib = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b)
    interp, spec, launches = both_modes(kc, lambda: chain32(kc, ia, ib, 32))
    assert launches == 1
    assert_planes(spec, interp, what="specialised vs interpreter")
    assert_planes(spec, orc.chain32(a, b, 32), what="specialised vs oracle")


@pytest.mark.parametrize("op", ["Add", "Subtract", "Multiply", "Divide", "Pow"])
@pytest.mark.parametrize("side", ["planes", "scalar_left", "scalar_right"])
def test_single_mix_every_op_and_operand_kind(kc, orc, op, side):
    h, w = 48, 200
    a = with_edge_cases(splitmix_plane(SEED_A, 0, h, w), 0)
    b = with_edge_cases(splitmix_plane(SEED_B, 0, h, w), 3)
    A, B = kc.SlotImage.from_planes([a]), kc.SlotImage.from_planes([b])
    c = np.float32(0.625)
    cplane = np.full((h, w), c, np.float32)

    def build():
        if side == "planes":
            return kc.mix_process(A, B, kc.MixType.parse(op))
        k = kc.resize_image(kc.value_process(float(c)), (w, h))
        return kc.mix_process(k, B, kc.MixType.parse(op)) if side == "scalar_left" else kc.mix_process(A, k, kc.MixType.parse(op))

    interp, spec, launches = both_modes(kc, build)
    assert launches == 1
    assert_planes(spec, interp, what="%s %s" % (op, side))  # the two device paths agree bit for bit, Pow included
    want = orc.mix_plane(op, a if side != "scalar_left" else cplane, b if side != "scalar_right" else cplane)
    assert_planes(spec, [want], ulp=1 if op == "Pow" else 0, what="%s %s vs oracle" % (op, side))


@pytest.mark.parametrize("shape", [(48, 200), (33, 130), (1, 7)])
@pytest.mark.parametrize("op", ["Add", "Subtract", "Multiply", "Divide", "Pow"])
@pytest.mark.parametrize("side", ["planes", "scalar_left", "scalar_right", "same_plane", "then_invert", "rgba"])
def test_single_mix_ahead_of_time_kernels(kc, orc, op, side, shape):
    """chain1.hip: a one-step program never reaches the interpreter; same bits as the interpreter, oracle-exact
    (Pow: 1 ulp).  then_invert = Mix(op) followed by 1 - x, which the host folds into ONE record (CH_*_INV)."""
    h, w = shape
    a = with_edge_cases(splitmix_plane(SEED_A, 0, h, w), 0)
    b = with_edge_cases(splitmix_plane(SEED_B, 0, h, w), 3)
    c = np.float32(0.625)
    cplane, ones = np.full((h, w), c, np.float32), np.ones((h, w), np.float32)
    mt = kc.MixType.parse(op)

    def build():
        A, B = kc.SlotImage.from_planes([a]), kc.SlotImage.from_planes([b])
        k = kc.resize_image(kc.value_process(float(c)), (w, h))
        if side == "planes":
            return kc.mix_process(A, B, mt)
        if side == "scalar_left":
            return kc.mix_process(k, B, mt)
        if side == "scalar_right":
            return kc.mix_process(A, k, mt)
        if side == "same_plane":
            return kc.mix_process(A, A, mt)
        if side == "then_invert":
            return kc.mix_process(kc.resize_image(kc.value_process(1.0), (w, h)), kc.mix_process(A, B, mt), kc.MixType.Subtract)
        return kc.mix_process(kc.SlotImage.from_planes([a, b, a, ones]), kc.SlotImage.from_planes([b, b, a, ones]), mt)

    # Divide / Pow followed by 1 - x stay two records: not a one-step program
    expect = 0 if side == "then_invert" and op in ("Divide", "Pow") else 1
    n0 = kc.stats_counter("chain1_launches")
    got = build().planes()
    assert kc.stats_counter("chain1_launches") - n0 == expect
    kc.set_option("chain1", 0)
    kc.set_specialize(0)
    try:
        interp = build().planes()
        assert kc.stats_counter("chain1_launches") - n0 == expect
    finally:
        kc.set_option("chain1", 1)
        kc.set_specialize(1)
    assert_planes(got, interp, what="%s %s ahead-of-time vs interpreter" % (op, side))
    m = lambda l, r: orc.mix_plane(op, l, r)
    want = {"planes": lambda: [m(a, b)], "scalar_left": lambda: [m(cplane, b)], "scalar_right": lambda: [m(a, cplane)],
            "same_plane": lambda: [m(a, a)], "then_invert": lambda: [orc.mix_plane("Subtract", ones, m(a, b))],
            "rgba": lambda: [m(a, b), m(b, b), m(a, a), ones]}[side]()
    if not (op == "Pow" and side == "then_invert"):  # 1 - x amplifies the one ulp Pow is allowed: interpreter-equal is the check there
        assert_planes(got, want, ulp=1 if op == "Pow" else 0, what="%s %s vs oracle" % (op, side))


def test_random_programs(kc, orc):
    rng = np.random.default_rng(20260402)
    h, w = 40, 72
    planes = [splitmix_plane(SEED_A + i, 0, h, w) * np.float32(0.75) + np.float32(0.125) for i in range(4)]
    imgs = [kc.SlotImage.from_planes([p]) for p in planes]
    ops = ["Add", "Subtract", "Multiply", "Divide"]
    for trial in range(24):
        n = int(rng.integers(1, 40))
        steps = [(ops[int(rng.integers(0, 4))], int(rng.integers(0, 6)), bool(rng.integers(0, 2))) for _ in range(n)]
        consts = [np.float32(rng.uniform(0.25, 1.0)) for _ in range(n)]  # a resized Value is clamped to [0, 1]

        def build():
            x = imgs[0]
            for (op, k, left), c in zip(steps, consts):
                o = imgs[k] if k < 4 else kc.resize_image(kc.value_process(float(c)), (w, h))
                x = kc.mix_process(o, x, kc.MixType.parse(op)) if left else kc.mix_process(x, o, kc.MixType.parse(op))
            return x

        want = planes[0]
        for (op, k, left), c in zip(steps, consts):
            o = planes[k] if k < 4 else np.full((h, w), c, np.float32)
            want = orc.mix_plane(op, o, want) if left else orc.mix_plane(op, want, o)
        interp, spec, launches = both_modes(kc, build)
        assert launches >= 1
        assert_planes(spec, interp, what="trial %d" % trial)
        assert_planes(spec, [want], what="trial %d vs oracle" % trial)


def test_constant_pow_constant_is_what_a_pixel_gets(kc, orc):
    """Pow is never folded on the host: Value ^ Value goes through the device's pow routine, so it equals the
    pixel value computed from planes holding the same numbers (in both device paths)."""
    h, w = 8, 64
    for lv, rv in ((0.33, 0.66), (1.7, 3.3), (0.001, 0.45), (7.5, -2.0), (0.0, 0.0), (-2.0, 3.0)):
        full = kc.mix_process(kc.SlotImage.from_planes([np.full((h, w), lv, np.float32)]),
                              kc.SlotImage.from_planes([np.full((h, w), rv, np.float32)]), kc.MixType.Pow).planes()[0]
        for mode in (0, 2):
            kc.set_specialize(mode)
            k = kc.mix_process(kc.value_process(lv), kc.value_process(rv), kc.MixType.Pow)
            assert k.size() == (1, 1)
            assert bit_equal(k.planes()[0], full[:1, :1]), (lv, rv, mode)


def test_background_compile_takes_over_after_repeated_sightings(kc, orc):
    h, w = 128, 256
    a = [splitmix_plane(SEED_A, c, h, w) for c in range(4)]
    b = [splitmix_plane(SEED_B, c, h, w) for c in range(4)]
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 0)
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(b)), 1)
    na = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
    nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
    n1 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
    n2 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply)))
    n3 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Divide)))
    lg.connect(na, n1, 0, 0)
    lg.connect(nb, n1, 0, 1)
    lg.connect(n1, n2, 0, 0)
    lg.connect(nb, n2, 0, 1)
    lg.connect(na, n3, 0, 0)
    lg.connect(n2, n3, 0, 1)
    want = [orc.mix_plane("Divide", a[c], orc.mix_plane("Multiply", orc.mix_plane("Subtract", a[c], b[c]), b[c])) for c in range(3)]
    want.append(np.ones((h, w), np.float32))
    kc.set_specialize(1, 3)  # compile on the third sighting
    try:
        s0 = kc.specialize_stats()
        for i in range(3):
            lg.connect(na, n1, 0, 0)  # re-dirties the chain
            assert_planes(lg.await_clean(n3).slot_data(n3, 0).image.planes(), want, what="evaluation %d" % i)
        assert kc.specialize_stats()["specialized_launches"] == s0["specialized_launches"]  # interpreter so far
        kc.specialize_wait()
        s1 = kc.specialize_stats()
        assert s1["kernels_compiled"] == s0["kernels_compiled"] + 1 and s1["compiles_pending"] == 0
        lg.connect(na, n1, 0, 0)
        assert_planes(lg.await_clean(n3).slot_data(n3, 0).image.planes(), want, what="specialised evaluation")
        assert kc.specialize_stats()["specialized_launches"] == s1["specialized_launches"] + 1
    finally:
        kc.set_specialize(1, 2)
