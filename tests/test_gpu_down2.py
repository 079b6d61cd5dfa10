"""GPU parity of resize_down2_kernel (csrc/down2.hip) -- down-sampling on both axes with windows of 4 taps or more each way, every wave
a job of its own -- against the CPU oracle and against the kernels it replaces (kc.set_option("down2", 0)), bit for bit,
through the C ABI.  Reference: image::imageops::resize (crate image 0.24.0) as called from src/shared.rs:159-199.
What is specific to this kernel: absent taps carry zero weights, which is exact only while the samples (vertical pass) and
the intermediate values (horizontal pass) are finite; waves that see an infinite or NaN value take an exact slow arm.  The
cases below put such values in the source, and make finite sources overflow in the vertical sums."""
import numpy as np
import pytest

from util import SEED_A, SEED_B, bit_equal, max_ulp, splitmix_plane

pytestmark = pytest.mark.gpu

CASES = [
    ("Lanczos3", (700, 300), (513, 219)),       # ratio 1.37: 9-10 taps, three columns per lane, four strips
    ("Lanczos3", (1000, 96), (250, 24)),        # ratio 4: 25 taps (resize_poly_kernel's ground: down2 = 2 only), three chunks
    ("CatmullRom", (1365, 260), (455, 87)),     # ratio 3: 13 taps, two columns per lane
    ("Gaussian", (3000, 120), (700, 28)),       # ratio 4.29: 27 taps, one column per lane
    ("Triangle", (4093, 205), (511, 26)),       # ratio 8: 17 taps, width not a multiple of 4
    ("Lanczos3", (260, 333), (190, 64)),        # 1.37 across, 5.2 down: 33 vertical taps, windows of four rows in four chunks
    ("CatmullRom", (130, 70), (61, 33)),        # one strip, partial last row group
    ("Gaussian", (64, 64), (16, 16)),
    ("CatmullRom", (1030, 70), (515, 35)),      # ratio 2: 8-9 taps
    ("Triangle", (1030, 70), (515, 35)),        # ratio 2: 4 taps, one weight quad per column
    ("Lanczos3", (300, 200), (290, 193)),       # ratio 1.03: 8 taps, windows of neighbouring rows almost coincide
    ("Gaussian", (200, 260), (48, 60)),         # ratio 4.2 on ONE strip, several chunks: the row-by-row job order has nothing to order
]


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    kc.set_option("poly2", 0)  # this module is about resize_down2_kernel and the kernels IT replaces (tests/test_gpu_poly2.py: the pair form)
    yield kc
    kc.set_option("down2", 1)
    kc.set_option("poly2", 1)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    return orc


def source(sh, sw, kind):
    p = splitmix_plane(SEED_A, 4, sh, sw) * np.float32(1.5) - np.float32(0.25)  # exercises the [0, 1] clamp
    if kind == "nonfinite":
        p[sh // 2, sw // 3:sw // 3 + 4] = [np.nan, np.inf, -np.inf, -0.0]
        p[0, 0] = -np.inf
        p[-1, -1] = np.inf
        p[sh // 3, -1] = np.nan
    elif kind == "overflow":
        # finite samples whose vertical sums overflow: the horizontal pass meets infinities the source did not hold
        p[sh // 2 - 3:sh // 2 + 3, sw // 2 - 2:sw // 2 + 2] = np.float32(3.0e38)
        p[1:5, 0:3] = np.float32(-3.2e38)
    elif kind == "denormal":
        p[::7, ::5] = np.float32(1e-42)
        p[3::7, 2::5] = np.float32(-0.0)
    return p


def resize(kc, p, dst, filt):
    return kc.resize_image(kc.SlotImage.from_planes([p]), dst, kc.ResizeFilter.parse(filt)).planes()[0]


@pytest.mark.parametrize("kind", ["finite", "nonfinite", "overflow", "denormal"])
@pytest.mark.parametrize("filt,src,dst", CASES)
def test_down2_equals_oracle_and_the_kernels_it_replaces(kc, orc, filt, src, dst, kind):
    (sw, sh), (dw, dh) = src, dst
    p = source(sh, sw, kind)
    want = orc.resize_plane(p, dw, dh, filt)
    got = {}
    try:
        for mode in (0, 1, 2, 3):
            # 3: mode 2 with the other job order (four strips of one row group per workgroup; by default only where the
            # windows span several chunks)
            kc.set_option("down2", min(mode, 2))
            kc.set_option("down2_by_rows", -1 if mode < 3 else (0 if kc.resize_down2_plan(sh, dh, kc.ResizeFilter.parse(filt))["nc"] > 1 else 1))
            n0 = kc.stats_counter("down2_launches")
            got[mode] = resize(kc, p, (dw, dh), filt)
            used = kc.stats_counter("down2_launches") - n0
            if mode == 0:
                assert used == 0
            if mode >= 2:
                f = kc.ResizeFilter.parse(filt)
                assert kc.resize_down2_plan(sh, dh, f)["nc"] and kc.resize_down2_plan(sw, dw, f)["tile_w"], "case does not fit the kernel"
                assert used == 1, "%s %s->%s should reach resize_down2_kernel" % (filt, src, dst)
    finally:
        kc.set_option("down2", 1)
        kc.set_option("down2_by_rows", -1)
    for mode in (0, 1, 2, 3):
        assert bit_equal(got[mode], want), "%s %s->%s %s down2=%d max ulp %s" % (filt, src, dst, kind, mode, max_ulp(got[mode], want))
    if kind in ("nonfinite", "overflow"):
        assert (~np.isfinite(want)).sum() + (want == 0).sum() + (want == 1).sum() > 0


def test_down2_rgba_planes_share_one_launch(kc, orc):
    planes = [source(240, 700, "nonfinite" if c == 2 else "finite") + np.float32(0.01 * c) for c in range(4)]
    kc.set_option("down2", 2)
    try:
        l0, n0 = kc.stats()["kernel_launches"], kc.stats_counter("down2_launches")
        got = kc.resize_image(kc.SlotImage.from_planes(planes), (513, 175), kc.ResizeFilter.Lanczos3)
        got.materialize()
        assert kc.stats()["kernel_launches"] - l0 == 1 and kc.stats_counter("down2_launches") - n0 == 1
    finally:
        kc.set_option("down2", 1)
    for c, g in enumerate(got.planes()):
        assert bit_equal(g, orc.resize_plane(planes[c], 513, 175, "Lanczos3")), c


def test_down2_full_size_non_integer_ratio(kc, orc):
    """VERDICT's case: Lanczos3 4096^2 -> 3000^2 (17 strips x 188 row tiles), against the oracle and with the counter."""
    p = source(4096, 4096, "nonfinite")
    n0 = kc.stats_counter("down2_launches")
    got = resize(kc, p, (3000, 3000), "Lanczos3")
    assert kc.stats_counter("down2_launches") - n0 == (1 if kc.get_option("down2") else 0)
    want = orc.resize_plane(p, 3000, 3000, "Lanczos3")
    assert bit_equal(got, want), max_ulp(got, want)


INTEGER_RATIO_CASES = [
    # integer ratios on the vertical axis (2, 4, 8) with every window length (Triangle 2 ages, CatmullRom 4, Gaussian / Lanczos3 6):
    # resize_poly_kernel's bands of 4, 8 and 12 rows (it picks the height by how many waves the image gives), widths that leave
    # partial strips, heights whose regular rows do not fill whole bands
    ("Lanczos3", (1000, 96), (250, 24)), ("Gaussian", (1024, 512), (128, 64)), ("Triangle", (2048, 256), (256, 32)),
    ("CatmullRom", (1200, 400), (300, 100)), ("Lanczos3", (516, 1032), (258, 516)), ("Gaussian", (4096, 160), (512, 20)),
    ("CatmullRom", (640, 1920), (320, 240)), ("Triangle", (808, 1616), (101, 202)), ("Lanczos3", (2000, 2000), (500, 500)),
    ("Gaussian", (2048, 2048), (256, 256)), ("Lanczos3", (4096, 4096), (1024, 1024)),
]


@pytest.mark.parametrize("kind", ["finite", "nonfinite"])
@pytest.mark.parametrize("filt,src,dst", INTEGER_RATIO_CASES)
def test_integer_ratio_down_sampling_equals_oracle(kc, orc, filt, src, dst, kind):
    (sw, sh), (dw, dh) = src, dst
    p = source(sh, sw, kind)
    want = orc.resize_plane(p, dw, dh, filt)
    got = resize(kc, p, (dw, dh), filt)
    assert bit_equal(got, want), "%s %s->%s %s max ulp %s" % (filt, src, dst, kind, max_ulp(got, want))
