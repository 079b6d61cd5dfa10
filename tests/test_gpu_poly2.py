"""GPU parity of resize_poly2_kernel (csrc/kernels.hip) -- down-sampling whose VERTICAL axis has an integer ratio (2, 4, 8), two
waves to a band's strip behind one s_barrier per four rows -- against the CPU oracle and against the kernels it replaces
(kc.set_option("poly2", 0): resize_poly_kernel / resize_down2_kernel), bit for bit, through the C ABI.
Reference: image::imageops::resize (crate image 0.24.0) as called from src/shared.rs:159-199.
What is specific to this kernel and covered below: strips wider than a wave (ratio 2: up to 128 output columns, the horizontal
pass one row at a time), strips of up to 32 / 64 columns (four / two rows per pass), an odd number of strips (the last one is
worked twice), a window whose right half is empty, bands of 4 / 8 / 12 rows, regular rows that do not fill the last band,
border rows as general tiles, a horizontal axis that is NOT an integer ratio, one to four planes per launch."""
import numpy as np
import pytest

from util import SEED_A, bit_equal, max_ulp, splitmix_plane

pytestmark = pytest.mark.gpu

CASES = [
    # (filter, source (w, h), destination (w, h))
    ("Gaussian", (4096, 512), (512, 64)),        # ratio 8, 6 ages: strips of <= 32 columns, one pass per four rows
    ("Lanczos3", (4096, 384), (1024, 96)),       # ratio 4: strips of 33..64 columns, two passes
    ("Lanczos3", (2048, 256), (1024, 128)),      # ratio 2: strips of 65..128 columns, four passes
    ("CatmullRom", (2048, 256), (256, 32)),      # 4 ages, ratio 8
    ("CatmullRom", (1200, 400), (300, 100)),     # 4 ages, ratio 4, 100 rows: the last band is short
    ("Gaussian", (1030, 140), (515, 70)),        # ratio 2, width not a multiple of 4 (Triangle / CatmullRom at ratio 2 have <= 8 taps: the register-tap kernels take them)
    ("Lanczos3", (640, 1920), (320, 240)),       # 2 across, 8 down
    ("CatmullRom", (808, 1616), (101, 202)),     # odd width, one or two strips
    ("Lanczos3", (700, 480), (513, 120)),        # 1.37 across (not an integer ratio), 4 down
    ("Gaussian", (3000, 240), (700, 30)),        # 4.29 across, 8 down: the regular rows barely exceed 16
    ("Lanczos3", (516, 1032), (258, 516)),
    ("Gaussian", (520, 256), (65, 32)),          # 65 columns: three strips (an odd count)
    ("Lanczos3", (2000, 2000), (500, 500)),
    ("Gaussian", (2048, 2048), (256, 256)),      # bands of 8 rows
    ("Lanczos3", (1024, 1024), (256, 256)),      # bands of 4 rows
]


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    kc.set_option("poly2_min_ratio", 2)  # by default the kernel takes ratio 8 only (where it is the fastest form); here: everything it can do
    yield kc
    kc.set_option("poly2", 1)
    kc.set_option("poly2_min_ratio", 8)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    return orc


def source(sh, sw, kind, c=0):
    p = splitmix_plane(SEED_A + c, 4, sh, sw) * np.float32(1.5) - np.float32(0.25)  # exercises the [0, 1] clamp
    if kind == "nonfinite":
        p[sh // 2, sw // 3:sw // 3 + 4] = [np.nan, np.inf, -np.inf, -0.0]
        p[0, 0] = -np.inf
        p[-1, -1] = np.inf
        p[sh // 3, -1] = np.nan
        p[sh // 2 - 3:sh // 2 + 3, sw // 2 - 2:sw // 2 + 2] = np.float32(3.0e38)  # finite samples whose vertical sums overflow
    return p


def resize(kc, planes, dst, filt):
    return kc.resize_image(kc.SlotImage.from_planes(planes), dst, kc.ResizeFilter.parse(filt)).planes()


@pytest.mark.parametrize("kind", ["finite", "nonfinite"])
@pytest.mark.parametrize("filt,src,dst", CASES)
def test_poly2_equals_oracle_and_the_kernels_it_replaces(kc, orc, filt, src, dst, kind):
    (sw, sh), (dw, dh) = src, dst
    p = source(sh, sw, kind)
    want = orc.resize_plane(p, dw, dh, filt)
    got = {}
    try:
        for mode in (0, 1):
            kc.set_option("poly2", mode)
            n0 = kc.stats_counter("poly2_launches")
            got[mode] = resize(kc, [p], (dw, dh), filt)[0]
            assert kc.stats_counter("poly2_launches") - n0 == mode, "%s %s->%s should%s reach resize_poly2_kernel" % (filt, src, dst, "" if mode else " not")
    finally:
        kc.set_option("poly2", 1)
    for mode in (0, 1):
        assert bit_equal(got[mode], want), "%s %s->%s %s poly2=%d max ulp %s" % (filt, src, dst, kind, mode, max_ulp(got[mode], want))


def test_poly2_rgba_planes_share_one_launch(kc, orc):
    planes = 4
    ps = [source(512, 1024, "nonfinite" if c == 2 else "finite", c) for c in range(planes)]
    l0, n0 = kc.stats()["kernel_launches"], kc.stats_counter("poly2_launches")
    img = kc.resize_image(kc.SlotImage.from_planes(ps), (256, 128), kc.ResizeFilter.Lanczos3)
    img.materialize()
    assert kc.stats()["kernel_launches"] - l0 == 1 and kc.stats_counter("poly2_launches") - n0 == 1
    for c, g in enumerate(img.planes()):
        assert bit_equal(g, orc.resize_plane(ps[c], 256, 128, "Lanczos3")), c


@pytest.mark.parametrize("order", [0, 1])
def test_poly2_min_ratio_option_and_repeat(kc, orc, order):
    """poly2_min_ratio keeps ratio-2 work on resize_down2_kernel when asked; the same launch twice gives the same bits (the ring
    and the barrier leave nothing behind)."""
    p = source(256, 2048, "finite")
    want = orc.resize_plane(p, 1024, 128, "Lanczos3")
    try:
        kc.set_option("poly2_min_ratio", 4 if order else 2)
        n0 = kc.stats_counter("poly2_launches")
        a = resize(kc, [p], (1024, 128), "Lanczos3")[0]
        b = resize(kc, [p], (1024, 128), "Lanczos3")[0]
        assert kc.stats_counter("poly2_launches") - n0 == (0 if order else 2)
    finally:
        kc.set_option("poly2_min_ratio", 2)
    assert bit_equal(a, want) and bit_equal(b, want)


def test_poly2_default_takes_ratio_8_with_wide_windows_only(kc, orc):
    """The shipped default (poly2_min_ratio 8, windows of 4 or 6 ages): Gaussian 8:1 goes through the kernel, Triangle 8:1 and
    Lanczos3 4:1 stay on resize_poly_kernel."""
    try:
        kc.set_option("poly2_min_ratio", 8)
        for filt, src, dst, used in (("Gaussian", (2048, 512), (256, 64), 1), ("Triangle", (2048, 512), (256, 64), 0), ("Lanczos3", (2048, 512), (512, 128), 0)):
            p = source(src[1], src[0], "finite")
            n0 = kc.stats_counter("poly2_launches")
            got = resize(kc, [p], dst, filt)[0]
            assert kc.stats_counter("poly2_launches") - n0 == used, (filt, src, dst)
            assert bit_equal(got, orc.resize_plane(p, dst[0], dst[1], filt)), (filt, src, dst)
    finally:
        kc.set_option("poly2_min_ratio", 2)
