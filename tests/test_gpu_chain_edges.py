"""Lazy-chain corner cases of the fused Mix kernel: program cuts (more than 64 steps, more than 4
distinct input planes), diamonds (both Mix operands lazy), the same plane on both sides, type
changes in the middle of a chain, constant folding, and plane-pool accounting.  Every case is
checked bit for bit against the oracle evaluated node by node."""
import numpy as np
import pytest

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
    return orc


def gray(kc, p):
    return kc.SlotImage.from_planes([p])


@pytest.mark.parametrize("wide", [0, 1])
def test_long_chain_is_cut_at_64_steps_and_4_planes(kc, orc, wide):
    """wide = 0: chains hold 4 input planes (what the interpreter handles); wide = 1 (the default): 16, on kernels compiled for the
    program (compiled at first sight here) -- 7 planes fit, only the 80-record limit cuts."""
    kc.set_option("wide", wide)
    kc.set_specialize(2)
    try:
        _long_chain(kc, orc, wide)
    finally:
        kc.set_option("wide", 1)
        kc.set_specialize(1)


def _long_chain(kc, orc, wide):
    h, w = 24, 40
    planes = [splitmix_plane(SEED_A + i, 0, h, w) * np.float32(0.5) + np.float32(0.25) for i in range(7)]
    imgs = [gray(kc, p) for p in planes]
    ops = ["Add", "Multiply", "Subtract", "Multiply", "Add"]
    x, want = imgs[0], planes[0]
    l0 = kc.stats()["kernel_launches"]
    for i in range(150):
        k = 1 + (i * 5 + i // 7) % 6          # walks over 6 other planes: forces > 4 distinct inputs
        op = ops[i % len(ops)]
        if i % 3 == 0:                          # operand on the left: x = planes[k] op x
            x = kc.mix_process(imgs[k], x, kc.MixType.parse(op))
            want = orc.mix_plane(op, planes[k], want)
        else:
            x = kc.mix_process(x, imgs[k], kc.MixType.parse(op))
            want = orc.mix_plane(op, want, planes[k])
    got = x.planes()
    launches = kc.stats()["kernel_launches"] - l0
    assert_planes(got, [want], what="150-step chain")
    if wide:
        assert launches == 2, launches               # 150 steps, 80 records per program
    else:
        assert 3 <= launches <= 150 // 3, launches   # cut every ~3-4 steps here (each restart spends one of the 4 input slots)


def test_diamond_both_operands_lazy(kc, orc):
    h, w = 33, 65
    a, b = splitmix_plane(SEED_A, 0, h, w) + np.float32(0.5), splitmix_plane(SEED_B, 0, h, w) + np.float32(0.5)
    A, B = gray(kc, a), gray(kc, b)
    x = kc.mix_process(A, B, kc.MixType.Add)            # lazy
    y1 = kc.mix_process(x, A, kc.MixType.Multiply)      # lazy, extends x
    y2 = kc.mix_process(x, B, kc.MixType.Subtract)      # lazy, extends x again (x is recomputed, never stored)
    z = kc.mix_process(y1, y2, kc.MixType.Divide)       # lazy x lazy
    xo = orc.mix_plane("Add", a, b)
    want = orc.mix_plane("Divide", orc.mix_plane("Multiply", xo, a), orc.mix_plane("Subtract", xo, b))
    assert_planes(z.planes(), [want], what="diamond")
    # the intermediates are still individually observable afterwards
    assert_planes(x.planes(), [xo])
    assert_planes(y2.planes(), [orc.mix_plane("Subtract", xo, b)])


def test_same_plane_on_both_sides_and_self_power(kc, orc):
    h, w = 16, 16
    a = splitmix_plane(SEED_A, 1, h, w)
    A = gray(kc, a)
    assert_planes(kc.mix_process(A, A, kc.MixType.Multiply).planes(), [orc.mix_plane("Multiply", a, a)])
    x = kc.mix_process(A, A, kc.MixType.Add)
    xx = kc.mix_process(x, x, kc.MixType.Subtract)     # lazy with itself
    assert_planes(xx.planes(), [orc.mix_plane("Subtract", orc.mix_plane("Add", a, a), orc.mix_plane("Add", a, a))])


def test_type_change_mid_chain_and_alpha(kc, orc):
    h, w = 20, 28
    a = [splitmix_plane(SEED_A, c, h, w) for c in range(4)]
    b = [splitmix_plane(SEED_B, c, h, w) for c in range(4)]
    g = splitmix_plane(SEED_B, 5, h, w)
    A, B, G = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b), gray(kc, g)
    x = kc.mix_process(A, B, kc.MixType.Multiply)                 # rgba, lazy
    y = kc.mix_process(G, x, kc.MixType.Add)                      # gray left: x averaged ((r+g)+b)/3, lazy chains merge
    z = kc.mix_process(y, x, kc.MixType.Subtract)                 # gray
    r = kc.mix_process(x, z, kc.MixType.Add)                      # rgba left, gray right broadcast
    xo = [orc.mix_plane("Multiply", a[c], b[c]) for c in range(3)]
    xg = orc.rgba_to_gray(*xo)
    yo = orc.mix_plane("Add", g, xg)
    zo = orc.mix_plane("Subtract", yo, xg)
    want = [orc.mix_plane("Add", xo[c], zo) for c in range(3)] + [np.ones_like(g)]
    assert r.is_rgba()
    assert_planes(r.planes(), want, what="type change")


def test_constant_folding_matches_device_arithmetic(kc, orc):
    one = np.ones((1, 1), np.float32)
    for op in ("Add", "Subtract", "Multiply", "Divide", "Pow"):
        for lv, rv in ((0.33, 0.66), (1.0, 3.0), (-2.0, 0.5), (0.0, 0.0), (7.5, -2.0)):
            img = kc.mix_process(kc.value_process(lv), kc.value_process(rv), kc.MixType.parse(op))
            want = orc.mix_plane(op, one * np.float32(lv), one * np.float32(rv))
            got = img.planes()[0]
            assert img.size() == (1, 1)
            assert bit_equal(got, want) or (op == "Pow" and abs(int(got.view(np.int32)[0, 0]) - int(want.view(np.int32)[0, 0])) <= 1), (op, lv, rv)


def test_pool_returns_to_baseline(kc):
    import gc
    gc.collect()
    kc.sync()
    base = kc.stats()["bytes_in_use"]
    h, w = 128, 192
    a = [splitmix_plane(SEED_A, c, h, w) for c in range(4)]
    A = kc.SlotImage.from_planes(a)
    assert kc.stats()["bytes_in_use"] == base + 4 * h * w * 4          # 192 * 4 B is already a multiple of 256
    x = kc.mix_process(A, A, kc.MixType.Add)
    assert kc.stats()["bytes_in_use"] == base + 4 * h * w * 4          # lazy: no bytes yet
    x.planes()
    assert kc.stats()["bytes_in_use"] == base + 7 * h * w * 4          # R, G, B materialised; alpha stays constant
    del x, A
    gc.collect()
    assert kc.stats()["bytes_in_use"] == base


def test_entry_points_are_thread_safe(kc, orc):
    """The reference runs process_node on one OS thread per ready node (src/engine.rs:288): the
    C ABI must tolerate concurrent callers.  8 threads build and evaluate their own graphs at once."""
    import threading
    h, w = 64, 96
    errors, results = [], {}

    def work(tid):
        try:
            a = [splitmix_plane(SEED_A + tid, c, h, w) for c in range(4)]
            b = [splitmix_plane(SEED_B + tid, c, h, w) for c in range(4)]
            tp = kc.TextureProcessor.new()
            for rep in range(5):
                lg = tp.new_live_graph()
                lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 0)
                lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(b)), 1)
                na = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
                nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
                prev = na
                for i in range(6):
                    n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add if i & 1 else kc.MixType.Multiply)))
                    lg.connect(prev, n, 0, 0)
                    lg.connect(nb, n, 0, 1)
                    prev = n
                small = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)).with_resize_policy(
                    kc.ResizePolicy.SpecificSize(kc.Size(w // 2, h // 2))))
                lg.connect(prev, small, 0, 0)
                results[(tid, rep)] = lg.await_clean(small).slot_data(small, 0).image.planes()
            want = [p.copy() for p in a[:3]]
            for i in range(6):
                want = [orc.mix_plane("Add" if i & 1 else "Multiply", want[c], b[c]) for c in range(3)]
            want = [orc.resize_plane(p, w // 2, h // 2, "Triangle") for p in want + [np.ones((h, w), np.float32)]]
            want = [orc.mix_plane("Add", p, np.zeros_like(p)) for p in want[:3]] + [np.ones((h // 2, w // 2), np.float32)]
            for rep in range(5):
                assert_planes(results[(tid, rep)], want, what="thread %d rep %d" % (tid, rep))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("filt", ["Nearest", "Triangle", "CatmullRom", "Gaussian", "Lanczos3"])
@pytest.mark.parametrize("small,big", [((16, 12), (128, 96)), ((110, 110), (128, 128)), ((33, 7), (131, 30)),
                                       ((64, 64), (48, 40)), ((1, 5), (9, 20))])
def test_resize_fused_into_the_consuming_chain(kc, orc, filt, small, big):
    """A resized input that only feeds Mix nodes is resampled inside the chain's own kernel
    (resize_chain_kernel) when its horizontal taps fit in registers; otherwise the plain resize
    kernel runs first.  Either way: bit-identical to the oracle and to the unfused evaluation."""
    (sw, sh), (w, h) = small, big
    a = [splitmix_plane(SEED_A, c, h, w) for c in range(4)]
    b = [splitmix_plane(SEED_B, c, sh, sw) * np.float32(1.4) - np.float32(0.2) for c in range(4)]
    if b[0].size >= 8:
        b[0].reshape(-1)[:4] = [np.nan, np.inf, -0.0, -np.inf]

    def run():
        tp = kc.TextureProcessor.new()
        lg = tp.new_live_graph()
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 0)
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(b)), 1)
        na = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
        nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
        pol = kc.ResizePolicy.SpecificSize(kc.Size(w, h))
        flt = kc.ResizeFilter.parse(filt)
        n1 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)).with_resize_policy(pol).with_resize_filter(flt))
        n2 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply)).with_resize_policy(pol).with_resize_filter(flt))
        n3 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)).with_resize_policy(pol).with_resize_filter(flt))
        lg.connect(na, n1, 0, 0)
        lg.connect(nb, n1, 0, 1)
        lg.connect(n1, n2, 0, 0)
        lg.connect(na, n2, 0, 1)
        lg.connect(nb, n3, 0, 0)          # the resampled plane on the LEFT this time
        lg.connect(n2, n3, 0, 1)
        l0 = kc.stats()["kernel_launches"]
        planes = lg.await_clean(n3).slot_data(n3, 0).image.planes()
        return planes, kc.stats()["kernel_launches"] - l0

    fused, launches = run()
    kc.set_fusion(False)
    try:
        unfused, _ = run()
    finally:
        kc.set_fusion(True)
    bu = [orc.resize_plane(p, w, h, filt) for p in b[:3]]
    want = []
    for c in range(3):
        x1 = orc.mix_plane("Add", a[c], bu[c])
        x2 = orc.mix_plane("Multiply", x1, a[c])
        want.append(orc.mix_plane("Subtract", bu[c], x2))
    want.append(np.ones((h, w), np.float32))
    assert_planes(fused, want, what="fused resize+chain %s" % filt)
    assert_planes(unfused, want, what="unfused %s" % filt)
    left, count, _ = orc.resize_taps(sw, w, filt)
    if int(count.max()) <= 4:
        assert launches == 1, launches          # resample + 3 Mix nodes x 3 channels: one kernel
    else:
        assert launches == 1 + 1, launches      # R, G, B resampled by one launch of the plain kernel, then one chain


@pytest.mark.parametrize("n_steps,op", [(20, "Add"), (3, "Divide")])
def test_resize_operand_falls_back_when_the_program_is_not_eligible(kc, orc, n_steps, op):
    """More than 16 steps, or a divide step, next to a resampled operand: the plain resize kernel
    runs first, then the ordinary chain kernel.  Same bits."""
    h, w, sh, sw = 40, 56, 10, 14
    a = splitmix_plane(SEED_A, 0, h, w) + np.float32(0.5)
    b = splitmix_plane(SEED_B, 0, sh, sw) * np.float32(0.9) + np.float32(0.05)
    A, B = gray(kc, a), gray(kc, b)
    bu = kc.resize_image(B, (w, h))
    x, want, bo = A, a, orc.resize_plane(b, w, h, "Triangle")
    l0 = kc.stats()["kernel_launches"]
    for i in range(n_steps):
        x = kc.mix_process(x, bu if i % 2 == 0 else A, kc.MixType.parse(op if i % 2 == 0 else "Multiply"))
        want = orc.mix_plane(op if i % 2 == 0 else "Multiply", want, bo if i % 2 == 0 else a)
    got = x.planes()
    assert kc.stats()["kernel_launches"] - l0 == 2
    assert_planes(got, [want], what="fallback %d %s" % (n_steps, op))


def test_constant_operand_materialised_after_the_chain_was_built(kc, orc):
    """A constant plane takes no input slot while it is a constant, but kc_plane_materialize (or a resize of
    it) turns it into a resident plane IN PLACE.  A lazy chain that already reads four planes then reads five
    when it is finally run: it must be split, not overflow the program's four input slots (found by UBSan under
    the graph fuzzer)."""
    h, w = 24, 40
    planes = [splitmix_plane(SEED_A + i, 0, h, w) for i in range(5)]
    imgs = [kc.SlotImage.from_planes([p]) for p in planes]
    const = kc.SlotImage.from_value((w, h), 0.375, False)
    x = kc.mix_process(imgs[0], imgs[1], kc.MixType.Add)
    x = kc.mix_process(x, imgs[2], kc.MixType.Multiply)
    x = kc.mix_process(x, const, kc.MixType.Subtract)
    x = kc.mix_process(x, imgs[3], kc.MixType.Add)      # four resident inputs + one constant: one program
    const.materialize()                                  # ... and now the constant is a fifth resident input
    want = orc.mix_plane("Add", planes[0], planes[1])
    want = orc.mix_plane("Multiply", want, planes[2])
    want = orc.mix_plane("Subtract", want, np.full((h, w), 0.375, np.float32))
    want = orc.mix_plane("Add", want, planes[3])
    l0 = kc.stats()["kernel_launches"]
    got = x.planes()[0]
    assert kc.stats()["kernel_launches"] - l0 == 2, "prefix and remainder"
    assert bit_equal(got, want)
