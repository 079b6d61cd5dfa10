"""GPU parity of the integer-ratio up-sampling kernels (csrc/upsample_chain.inc: upsample_kernel, upsample_chain_kernel
and the run-time specialised form) against the CPU oracle, and against the general kernels they replace
(kc_set_resize_mode(4)) -- bit for bit, through the C ABI.  Reference: src/shared.rs:159-199 feeding
src/node/mix.rs:136-192."""
import numpy as np
import pytest

from util import SEED_A, SEED_B, assert_planes, bit_equal, max_ulp, splitmix_plane, synthetic_rgba

pytestmark = pytest.mark.gpu

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


def nasty(p):
    """Clamp-exercising range plus non-finite samples at the corners, in the middle and in the first window."""
    p = p * np.float32(1.5) - np.float32(0.25)
    h, w = p.shape
    f = p.reshape(-1)
    if f.size >= 8:
        f[:4] = [np.nan, np.inf, -np.inf, -0.0]
        p[h // 2, w // 2] = np.inf
        p[-1, -1] = -np.inf
        p[-1, 0] = np.nan
        p[0, -1] = 5.877e-39  # denormal
    return p


@pytest.mark.parametrize("filt", FILTERS)
@pytest.mark.parametrize("src,dst", [
    ((16, 16), (128, 128)),      # 8x both ways, one tile
    ((64, 48), (256, 48)),       # 4x across, 1x down (the vertical table is the identity-like one of ratio 1)
    ((37, 21), (148, 63)),       # 4x across, 3x down (chunk 1), odd source width: partial source quad
    ((130, 50), (2080, 100)),    # 16x across: three 1024-wide tiles, 2x down
    ((9, 300), (108, 600)),      # 12x across (not a power of two), many row tiles
    ((3, 3), (48, 6)),
    ((300, 7), (1200, 448)),     # 64x down the rows: chunks of 8 inside one window
    ((5, 5), (20, 40)),
    ((64, 40), (128, 80)),       # 2x both ways: "half quads" (columns 0, 1 and 2, 3 of a thread have windows one sample apart)
    ((130, 50), (260, 200)),     # 2x across, 4x down
    ((1100, 16), (2200, 32)),    # 2x: three 1024-wide tiles
    ((10, 10), (20, 20)),        # 2x, the smallest sizes the half quads take
    ((7, 9), (14, 18)),          # 2x with a width that is not a multiple of 4: the general kernels
])
def test_plain_upsample_bit_exact(kc, orc, filt, src, dst):
    (sw, sh), (dw, dh) = src, dst
    f = kc.ResizeFilter.parse(filt)
    p = nasty(splitmix_plane(SEED_A, 0, sh, sw))
    ph, pv = kc.resize_upsample_plan(sw, dw, f), kc.resize_upsample_plan(sh, dh, f)
    expect_new = ph is not None and pv is not None and ph["taps"] == pv["taps"] and (
        ph["ratio"] % 4 == 0 or (ph["ratio"] == 2 and dw % 4 == 0 and sw >= 8))
    n0 = kc.stats_counter("upsample_launches")
    got = kc.resize_image(kc.SlotImage.from_planes([p]), (dw, dh), f).planes()[0]
    assert (kc.stats_counter("upsample_launches") - n0 == 1) == expect_new
    want = orc.resize_plane(p, dw, dh, filt)
    assert bit_equal(got, want), "%s %s->%s max ulp %s" % (filt, src, dst, max_ulp(got, want))
    kc.set_resize_mode(4)
    try:
        n0 = kc.stats_counter("upsample_launches")
        old = kc.resize_image(kc.SlotImage.from_planes([p]), (dw, dh), f).planes()[0]
        assert kc.stats_counter("upsample_launches") == n0
    finally:
        kc.set_resize_mode(0)
    assert bit_equal(got, old)


def test_the_cases_above_do_reach_the_new_kernel(kc):
    for (sw, dw), (sh, dh) in (((16, 128), (16, 128)), ((64, 256), (48, 48)), ((130, 2080), (50, 100)), ((9, 108), (300, 600))):
        for filt in FILTERS:
            f = kc.ResizeFilter.parse(filt)
            ph, pv = kc.resize_upsample_plan(sw, dw, f), kc.resize_upsample_plan(sh, dh, f)
            assert ph is not None and pv is not None and ph["taps"] == pv["taps"], (filt, sw, dw, sh, dh)


def test_rgba_planes_share_one_launch(kc, orc):
    planes = [nasty(splitmix_plane(SEED_B, c, 96, 130)) for c in range(4)]
    l0, n0 = kc.stats()["kernel_launches"], kc.stats_counter("upsample_launches")
    got = kc.resize_image(kc.SlotImage.from_planes(planes), (1040, 768), kc.ResizeFilter.Triangle)
    got.materialize()
    assert kc.stats()["kernel_launches"] - l0 == 1 and kc.stats_counter("upsample_launches") - n0 == 1
    for c, g in enumerate(got.planes()):
        assert bit_equal(g, orc.resize_plane(planes[c], 1040, 768, "Triangle")), c


def test_512_to_4096_triangle_full_size(kc, orc):
    """BASELINE config #2's resample on its own: 512^2 -> 4096^2, checked on three row strips against the oracle."""
    p = splitmix_plane(SEED_B, 0, 512, 512)
    got = kc.resize_image(kc.SlotImage.from_planes([p]), (4096, 4096), kc.ResizeFilter.Triangle).planes()[0]
    want = orc.resize_plane(p, 4096, 4096, "Triangle")
    assert bit_equal(got, want), max_ulp(got, want)


def _mix(kc, op, left, right):
    return kc.mix_process(left, right, getattr(kc.MixType, op))


@pytest.mark.parametrize("specialize", [0, 2])
@pytest.mark.parametrize("filt", ["Triangle", "Nearest"])
@pytest.mark.parametrize("small,big", [((16, 16), (128, 128)), ((130, 12), (2080, 96)), ((37, 5), (148, 15)), ((64, 64), (256, 512)),
                                       ((64, 40), (128, 80)), ((1100, 16), (2200, 64))])  # the last two: ratio 2 across
def test_fused_upsample_chain_bit_exact(kc, orc, small, big, filt, specialize):
    """The resampled operand feeds a Mix chain inside one launch (config #2's shape): K = 2 resident + resampled."""
    (sw, sh), (dw, dh) = small, big
    f = kc.ResizeFilter.parse(filt)
    a = synthetic_rgba(SEED_A, dh, dw)
    b = [nasty(q) for q in synthetic_rgba(SEED_B, sh, sw)]
    kc.set_specialize(specialize)
    try:
        ia, ib = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b)
        up = kc.resize_image(ib, (dw, dh), f)
        n0 = kc.stats_counter("upsample_chain_launches")
        l0 = kc.stats()["kernel_launches"]
        n1 = _mix(kc, "Add", ia, up)
        n2 = _mix(kc, "Multiply", n1, ia)
        n3 = _mix(kc, "Subtract", n2, up)
        got = n3.planes()
        assert kc.stats()["kernel_launches"] - l0 == 1
        assert kc.stats_counter("upsample_chain_launches") - n0 == 1
    finally:
        kc.set_specialize(1)
    bu = [orc.resize_plane(q, dw, dh, filt) for q in b[:3]]
    want = []
    for c in range(3):
        t = orc.mix_plane("Add", a[c], bu[c])
        t = orc.mix_plane("Multiply", t, a[c])
        want.append(orc.mix_plane("Subtract", t, bu[c]))
    want.append(np.ones((dh, dw), np.float32))
    assert_planes(got, want, ulp=0, what="fused upsample + chain")


@pytest.mark.parametrize("n_resident", [0, 1, 2, 3])
def test_fused_upsample_chain_every_input_count(kc, orc, n_resident):
    """K = 1 .. 4 inputs: the resampled operand alone (with constants), and next to one, two and three resident planes."""
    sw, sh, dw, dh = 24, 10, 192, 40
    b = nasty(splitmix_plane(SEED_B, 1, sh, sw))
    res = [splitmix_plane(SEED_A, 5 + i, dh, dw) for i in range(n_resident)]
    up = kc.resize_image(kc.SlotImage.from_planes([b]), (dw, dh), kc.ResizeFilter.Triangle)
    n0 = kc.stats_counter("upsample_chain_launches")
    x = _mix(kc, "Multiply", up, kc.SlotImage.from_value((dw, dh), 0.75, False))
    bu = orc.resize_plane(b, dw, dh, "Triangle")
    want = orc.mix_plane("Multiply", bu, np.full((dh, dw), 0.75, np.float32))
    for i, r in enumerate(res):
        op = ["Add", "Subtract", "Multiply"][i]
        x = _mix(kc, op, x, kc.SlotImage.from_planes([r]))
        want = orc.mix_plane(op, want, r)
    x = _mix(kc, "Subtract", up, x)
    want = orc.mix_plane("Subtract", bu, want)
    got = x.planes()
    assert kc.stats_counter("upsample_chain_launches") - n0 == 1
    assert_planes(got, [want], ulp=0, what="K = %d" % (n_resident + 1))


def test_config2_shape_matches_general_kernel_and_oracle(kc, orc):
    """512^2 -> 4096^2 x 3-node blend chain is too slow for the oracle at full size in a unit test budget only on the
    CPU side; 128^2 -> 1024^2 has the same tiles (1024 wide), the same phases and borders."""
    a = synthetic_rgba(SEED_A, 1024, 1024)
    b = synthetic_rgba(SEED_B, 128, 128)

    def run():
        ia, ib = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b)
        up = kc.resize_image(ib, (1024, 1024), kc.ResizeFilter.Triangle)
        return _mix(kc, "Subtract", _mix(kc, "Multiply", _mix(kc, "Add", ia, up), ia), up).planes()

    got = run()
    kc.set_resize_mode(4)
    try:
        old = run()
    finally:
        kc.set_resize_mode(0)
    for c in range(3):
        assert bit_equal(got[c], old[c]), c
        bu = orc.resize_plane(b[c], 1024, 1024, "Triangle")
        want = orc.mix_plane("Subtract", orc.mix_plane("Multiply", orc.mix_plane("Add", a[c], bu), a[c]), bu)
        assert bit_equal(got[c], want), c
