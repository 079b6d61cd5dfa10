"""GPU parity, level 2: each HIP kernel against the CPU oracle at f32 on seeded synthetic planes
(with IEEE edge cases spliced in), through the C ABI operator entry points."""
import os

import numpy as np
import pytest

from util import (SEED_A, SEED_B, assert_planes, bit_equal, max_ulp, splitmix_plane, synthetic_rgba,
                  with_edge_cases)

pytestmark = pytest.mark.gpu

OPS = ["Add", "Subtract", "Multiply", "Divide", "Pow"]
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


def _ulp(op):
    return 1 if op == "Pow" else 0


@pytest.mark.parametrize("shape", [(64, 64), (37, 101), (1, 1), (3, 1), (1, 7), (256, 260)])
@pytest.mark.parametrize("op", OPS)
def test_mix_gray(kc, orc, op, shape):
    h, w = shape
    a = with_edge_cases(splitmix_plane(SEED_A, 0, h, w), 1)
    b = with_edge_cases(splitmix_plane(SEED_B, 0, h, w), 3)
    got = kc.mix_process(kc.SlotImage.from_planes([a]), kc.SlotImage.from_planes([b]), kc.MixType.parse(op))
    assert not got.is_rgba()
    assert_planes(got.planes(), [orc.mix_plane(op, a, b)], ulp=_ulp(op), what=op)


@pytest.mark.parametrize("fusion", [True, False])
@pytest.mark.parametrize("op", OPS)
def test_mix_rgba_alpha_is_one_and_input_alpha_ignored(kc, orc, op, fusion):
    h, w = 96, 130
    a = [with_edge_cases(p, 1) for p in synthetic_rgba(SEED_A, h, w)]
    b = [with_edge_cases(p, 2) for p in synthetic_rgba(SEED_B, h, w)]
    kc.set_fusion(fusion)
    try:
        got = kc.mix_process(kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b), kc.MixType.parse(op))
        planes = got.planes()
    finally:
        kc.set_fusion(True)
    want = [orc.mix_plane(op, a[c], b[c]) for c in range(3)] + [np.ones((h, w), np.float32)]
    assert_planes(planes, want, ulp=_ulp(op), what=op)


def test_mix_missing_inputs_and_type_matching(kc, orc):
    h, w = 40, 48
    a = synthetic_rgba(SEED_A, h, w)
    g = splitmix_plane(SEED_B, 0, h, w)
    rgba, gray = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes([g])
    # only left: right = zeros of left's type (mix.rs:64)
    got = kc.mix_process(rgba, None, kc.MixType.Add).planes()
    assert_planes(got, [orc.mix_plane("Add", a[c], np.zeros_like(g)) for c in range(3)] + [np.ones_like(g)])
    # only right: left = zeros, Subtract gives 0 - r (mix.rs:69-76)
    got = kc.mix_process(None, gray, kc.MixType.Subtract).planes()
    assert_planes(got, [orc.mix_plane("Subtract", np.zeros_like(g), g)])
    # neither: 1x1 gray 0.0 (mix.rs:77-83)
    img = kc.mix_process(None, None, kc.MixType.Add)
    assert img.size() == (1, 1) and not img.is_rgba() and img.planes()[0][0, 0] == 0.0
    # gray left, rgba right: right averaged to gray ((r+g)+b)/3 first, output gray (mix.rs:57-62)
    got = kc.mix_process(gray, rgba, kc.MixType.Multiply)
    assert not got.is_rgba()
    assert_planes(got.planes(), [orc.mix_plane("Multiply", g, orc.rgba_to_gray(*a[:3]))])
    # rgba left, gray right: right broadcast to [p, p, p, 1]
    got = kc.mix_process(rgba, gray, kc.MixType.Divide)
    assert_planes(got.planes(), [orc.mix_plane("Divide", a[c], g) for c in range(3)] + [np.ones_like(g)])


def test_as_type_and_from_value(kc, orc):
    h, w = 33, 77
    a = [with_edge_cases(p, 1) for p in synthetic_rgba(SEED_A, h, w)]
    img = kc.SlotImage.from_planes(a)
    assert_planes(img.as_type(False).planes(), [orc.rgba_to_gray(*a[:3])])
    g = kc.SlotImage.from_planes([a[0]]).as_type(True).planes()
    assert_planes(g, [a[0], a[0], a[0], np.ones_like(a[0])])
    v = kc.SlotImage.from_value((5, 3), 0.25, True).planes()
    assert [float(p[0, 0]) for p in v] == [0.25, 0.25, 0.25, 1.0] and v[0].shape == (3, 5)


@pytest.mark.parametrize("shape", [(16, 16), (9, 13)])
def test_fused_chain_equals_unfused_and_oracle(kc, orc, shape):
    """The 32-node linear graph of SURVEY.md 8(d) config #3 as direct operator calls."""
    h, w = shape
    a, b = synthetic_rgba(SEED_A, h, w), synthetic_rgba(SEED_B, h, w)
    want = orc.chain32(a, b, 32)

    def run():
        x = kc.SlotImage.from_planes(a)
        bb = kc.SlotImage.from_planes(b)
        white = kc.combine_rgba_process([kc.value_process(1.0)] * 3 + [None])
        for i in range(1, 33):
            if i & 1:
                x = kc.mix_process(x, bb, kc.MixType.Multiply if (i >> 1) & 1 else kc.MixType.Add)
            else:
                x = kc.mix_process(kc.resize_image(white, (w, h)), x, kc.MixType.Subtract)
        return x.planes()

    l0 = kc.stats()["kernel_launches"]
    fused = run()
    l1 = kc.stats()["kernel_launches"]
    kc.set_fusion(False)
    try:
        unfused = run()
    finally:
        kc.set_fusion(True)
    l2 = kc.stats()["kernel_launches"]
    assert_planes(fused, want, what="fused vs oracle")
    assert_planes(unfused, want, what="unfused vs oracle")
    assert l1 - l0 == 1, "the whole chain must be one kernel launch"
    assert l2 - l1 == 32


@pytest.mark.parametrize("filt", FILTERS)
@pytest.mark.parametrize("src,dst", [((16, 16), (128, 128)), ((110, 110), (128, 128)), ((64, 48), (17, 23)),
                                     ((256, 256), (10, 10)), ((10, 10), (20, 20)), ((1, 1), (9, 5)),
                                     ((5, 1), (3, 7)), ((300, 2), (2, 300))])
def test_resize_bit_exact(kc, orc, filt, src, dst):
    (sw, sh), (dw, dh) = src, dst
    p = splitmix_plane(SEED_A, 0, sh, sw) * np.float32(1.5) - np.float32(0.25)  # exercises the [0,1] clamp
    if p.size >= 16:
        p.reshape(-1)[:4] = [np.nan, np.inf, -np.inf, -0.0]
    got = kc.resize_image(kc.SlotImage.from_planes([p]), (dw, dh), kc.ResizeFilter.parse(filt)).planes()[0]
    want = orc.resize_plane(p, dw, dh, filt)
    assert bit_equal(got, want), "%s %s->%s max ulp %s" % (filt, src, dst, max_ulp(got, want))


# Several tiles in both directions, widths that are not multiples of 4, every kernel form:
# register taps (<= 8 per output), LDS tap table (wide windows), up in one axis and down in the other.
@pytest.mark.parametrize("filt,src,dst", [
    ("Triangle", (130, 50), (2050, 90)),      # two 1024-wide tiles, partial last quad
    ("Lanczos3", (130, 50), (1031, 41)),      # 8 register taps
    ("Triangle", (1030, 70), (515, 35)),      # down 2x: 5 register taps
    ("CatmullRom", (1030, 70), (515, 35)),    # down 2x: 9 taps -> LDS tap table
    ("Triangle", (520, 133), (65, 17)),       # down 8x, three column tiles of the wide kernel
    ("Lanczos3", (333, 520), (41, 65)),       # down ~8x, 49 taps per axis
    ("Gaussian", (64, 512), (512, 64)),       # up horizontally, down vertically
    ("Triangle", (512, 64), (64, 512)),       # down horizontally, up vertically
    ("Nearest", (1000, 9), (37, 1)),
    ("Triangle", (4096, 3), (3000, 2)),
])
def test_resize_bit_exact_many_tiles(kc, orc, filt, src, dst):
    (sw, sh), (dw, dh) = src, dst
    p = splitmix_plane(SEED_B, 2, sh, sw) * np.float32(1.5) - np.float32(0.25)
    p.reshape(-1)[5:9] = [np.nan, np.inf, -np.inf, -0.0]
    p[-1, -1] = np.inf  # last source column / row: read by the final, partial 4-column group
    got = kc.resize_image(kc.SlotImage.from_planes([p]), (dw, dh), kc.ResizeFilter.parse(filt)).planes()[0]
    want = orc.resize_plane(p, dw, dh, filt)
    assert bit_equal(got, want), "%s %s->%s max ulp %s" % (filt, src, dst, max_ulp(got, want))


# Integer-ratio down-sampling (resize_poly_kernel): every filter family (2, 4 and 6 "ages"), ratios 2, 4 and 8, different
# ratios on the two axes, several bands and strips, heights whose regular rows fill neither a band nor a group of four,
# widths that are not multiples of 4, and non-finite samples that must stay inside their windows.
@pytest.mark.parametrize("filt,src,dst", [
    ("Triangle", (1024, 1024), (256, 256)),
    ("CatmullRom", (1024, 1024), (256, 256)),
    ("Lanczos3", (1024, 1024), (256, 256)),
    ("Gaussian", (1024, 1016), (512, 508)),
    ("Lanczos3", (1000, 2064), (125, 258)),
    ("Triangle", (2050, 520), (1025, 65)),       # ratio 2 across, 8 down
    ("CatmullRom", (516, 2056), (129, 257)),     # ratio 4 across, 8 down
    ("Gaussian", (2048, 200), (512, 50)),
    ("Lanczos3", (4093, 236), (1025, 118)),      # not an integer ratio across: general horizontal taps, regular vertical ones
    ("Lanczos3", (256, 256), (64, 64)),
])
def test_resize_integer_ratio_bands(kc, orc, filt, src, dst):
    (sw, sh), (dw, dh) = src, dst
    p = splitmix_plane(SEED_A, 3, sh, sw) * np.float32(1.5) - np.float32(0.25)
    p[sh // 2, sw // 3:sw // 3 + 4] = [np.nan, np.inf, -np.inf, -0.0]
    p[0, 0] = -np.inf
    p[-1, -1] = np.inf
    got = kc.resize_image(kc.SlotImage.from_planes([p]), (dw, dh), kc.ResizeFilter.parse(filt)).planes()[0]
    want = orc.resize_plane(p, dw, dh, filt)
    assert bit_equal(got, want), "%s %s->%s max ulp %s" % (filt, src, dst, max_ulp(got, want))
    assert np.isnan(got).sum() == np.isnan(want).sum() > 0


def test_resize_integer_ratio_rgba_one_launch(kc, orc):
    planes = [splitmix_plane(SEED_B, c, 1040, 772) for c in range(4)]
    l0 = kc.stats()["kernel_launches"]
    got = kc.resize_image(kc.SlotImage.from_planes(planes), (193, 260), kc.ResizeFilter.Lanczos3)
    got.materialize()
    assert kc.stats()["kernel_launches"] - l0 == 1
    for c, g in enumerate(got.planes()):
        assert bit_equal(g, orc.resize_plane(planes[c], 193, 260, "Lanczos3")), c


@pytest.mark.parametrize("fusion", [True, False])
def test_resize_rgba_planes_share_one_launch(kc, orc, fusion):
    """The planes of an image are resampled by one launch (blockIdx.z = plane); aliased planes once."""
    h, w = 40, 52
    a = synthetic_rgba(SEED_A, h, w)
    kc.set_fusion(fusion)
    try:
        l0 = kc.stats()["kernel_launches"]
        got = kc.resize_image(kc.SlotImage.from_planes(a), (130, 100), kc.ResizeFilter.CatmullRom).planes()
        l1 = kc.stats()["kernel_launches"]
        gray = kc.SlotImage.from_planes([a[0]]).as_type(True)  # [p, p, p, ones]
        got_g = kc.resize_image(gray, (130, 100)).planes()
        l2 = kc.stats()["kernel_launches"]
    finally:
        kc.set_fusion(True)
    assert_planes(got, [orc.resize_plane(p, 130, 100, "CatmullRom") for p in a], what="rgba resize")
    assert l1 - l0 == 1
    want = orc.resize_plane(a[0], 130, 100, "Triangle")
    assert_planes(got_g, [want, want, want, np.ones((100, 130), np.float32)], what="aliased planes")
    # p once; the ones plane is resampled like any other plane (sum of f32 weights, as the reference
    # does): one fill to make it resident, then ONE launch for both
    assert l2 - l1 == 2


@pytest.mark.parametrize("src,dst", [((8192, 8), (12, 5)), ((8, 8192), (5, 12)), ((3000, 3000), (2, 2))])
def test_resize_two_pass_fallback_matches(kc, orc, src, dst):
    # down-sampling windows (about 4096 source columns / rows per output) too large for the smallest
    # LDS tile, horizontally, vertically and both: the vertical + horizontal kernels run instead
    (sw, sh), (dw, dh) = src, dst
    p = splitmix_plane(SEED_B, 1, sh, sw)
    l0 = kc.stats()["kernel_launches"]
    got = kc.resize_image(kc.SlotImage.from_planes([p]), (dw, dh), kc.ResizeFilter.Lanczos3).planes()[0]
    assert kc.stats()["kernel_launches"] - l0 == 2, "expected the vertical + horizontal kernels"
    assert bit_equal(got, orc.resize_plane(p, dw, dh, "Lanczos3"))


def test_resize_source_in_wrapped_memory_with_tight_pitch(kc, orc):
    """16-byte loads may touch the pitch padding of a row, never the next allocation: a caller-owned
    source whose pitch is exactly 16 * ceil(w / 4) bytes, last row at the end of its buffer."""
    import ctypes as C
    import torch
    from kanter_core_amd import _lib
    L = _lib.load()
    h, w = 19, 10
    pitch_f = 12
    t = torch.full((h, pitch_f), float("nan"), device="cuda")
    p = splitmix_plane(SEED_A, 3, h, w)
    t[:, :w] = torch.from_numpy(p).cuda()
    torch.cuda.synchronize()
    plane, img = C.c_void_p(), C.c_void_p()
    assert L.kc_plane_wrap(t.data_ptr(), w, h, pitch_f * 4, C.byref(plane)) == 0
    L.kc_image_gray(plane, C.byref(img))
    src = kc.SlotImage(img.value)
    for dst, filt in (((37, 40), "Triangle"), ((5, 4), "Lanczos3"), ((10, 38), "CatmullRom")):
        got = kc.resize_image(src, dst, kc.ResizeFilter.parse(filt)).planes()[0]
        assert bit_equal(got, orc.resize_plane(p, dst[0], dst[1], filt)), (dst, filt)
    L.kc_plane_release(plane)
    assert L.kc_plane_wrap(t.data_ptr(), w, h, 40, C.byref(plane)) != 0  # pitch not a multiple of 16 bytes


def test_resize_1x1_value_broadcast_is_clamped_constant(kc, orc):
    for v in (0.33, 1.0, 7.0, -3.0, float("nan"), -0.0):
        img = kc.resize_image(kc.value_process(v), (6, 4))
        want = orc.resize_plane(np.full((1, 1), v, np.float32), 6, 4, "Triangle")
        assert bit_equal(img.planes()[0], want), v


@pytest.mark.parametrize("shape", [(64, 64), (31, 45), (1, 1), (2, 5), (256, 256)])
def test_height_to_normal(kc, orc, shape):
    h, w = shape
    p = splitmix_plane(SEED_A, 2, h, w)
    if p.size >= 64:  # IEEE edge cases in the height field: every lane must still agree with the oracle
        p.reshape(-1)[3:60:8] = [np.nan, np.inf, -np.inf, 1e30, -1e30, 0.0, -0.0, 1e-40]
    got = kc.height_to_normal_process(kc.SlotImage.from_planes([p]))
    nx, ny, nz = orc.height_to_normal(p)
    assert_planes(got.planes(), [nx, ny, nz, np.ones_like(p)], ulp=0, what="h2n")
    assert kc.height_to_normal_process(None) is None
    assert kc.height_to_normal_process(kc.SlotImage.from_value((2, 2), 0.0, True)) is None


def test_height_to_normal_across_magnitudes(kc, orc):
    """Height steps from 2^-60 to 2^30, flat runs, and random mantissas: both the shared-denominator
    division path (steps of 0 or 2^-40 .. 2^7) and the general one, pixel by pixel against the oracle."""
    rng = np.random.default_rng(7)
    h, w = 384, 512
    step = np.ldexp(rng.random((h, w)) + 0.5, rng.integers(-60, 31, (h, w))).astype(np.float32)
    step *= rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), (h, w), p=[0.4, 0.2, 0.4])
    p = np.cumsum(step, axis=1, dtype=np.float32)
    p[:64] = np.ldexp(rng.random((64, w)), rng.integers(-45, -35, (64, w))).astype(np.float32)  # around the 2^-40 bound
    p[64:128] = np.ldexp(rng.random((64, w)), rng.integers(4, 11, (64, w))).astype(np.float32)  # around the 2^7 bound
    p[128:160] = 0.25  # flat: every step is +0
    p[160, :8] = [np.nan, np.inf, -np.inf, -0.0, 0.0, 1e-45, -1e-45, 3e38]
    got = kc.height_to_normal_process(kc.SlotImage.from_planes([p]))
    nx, ny, nz = orc.height_to_normal(p)
    assert_planes(got.planes(), [nx, ny, nz, np.ones_like(p)], ulp=0, what="h2n magnitudes")


@pytest.mark.parametrize("srgb", [False, True])
@pytest.mark.parametrize("shape", [(32, 32), (7, 13)])
def test_to_u8(kc, orc, shape, srgb):
    h, w = shape
    planes = [with_edge_cases(p * np.float32(1.2) - np.float32(0.1), 1) for p in synthetic_rgba(SEED_B, h, w)]
    got = kc.SlotImage.from_planes(planes).to_u8(srgb)
    want = orc.to_u8(orc.Image(planes), srgb)
    # sRGB too: the export is evaluated as a step function whose thresholds come from libm's powf (tools/gen_srgb_thresholds.c)
    assert np.array_equal(got, want)
    gg = kc.SlotImage.from_planes(planes[:1]).to_u8(srgb)
    wg = orc.to_u8(orc.Image(planes[:1]), srgb)
    assert np.array_equal(gg, wg)


def test_to_u8_srgb_at_every_threshold(kc, orc):
    """Every float within 8 ulps of each of the 255 level thresholds, 2^22 other floats of [0, 1] and the special values."""
    import re
    inc = open(os.path.join(os.path.dirname(__file__), "..", "kanter_core_amd", "csrc", "srgb_thresholds.inc")).read()
    T = np.array([int(x, 16) for x in re.findall(r"0x([0-9a-f]{8})u", inc)], dtype=np.int64)
    assert T.size == 256 and (np.diff(T[1:]) > 0).all()
    near = (T[1:, None] + np.arange(-8, 9)[None, :]).reshape(-1)
    rng = np.random.default_rng(11)
    rand = rng.integers(0, 0x3F800001, 1 << 22)
    special = np.array([0, 0x80000000, 0x3F800000, 0x3F800001, 0x7F800000, 0xFF800000, 0x7FC00000, 1, 0x00800000, 0xBF800000], dtype=np.int64)
    bits = np.concatenate([near, rand, special]).astype(np.uint32)
    pad = (-bits.size) % 1024
    bits = np.concatenate([bits, np.zeros(pad, np.uint32)])
    x = bits.view(np.float32).reshape(-1, 1024)
    got = kc.SlotImage.from_planes([x]).to_u8(True)
    want = orc.to_u8(orc.Image([x]), True)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
def test_from_u8(kc, orc, channels):
    rng = np.random.default_rng(7)
    px = rng.integers(0, 256, (19, 23, channels), dtype=np.uint8)
    got = kc.SlotImage.from_u8(px)
    assert got.is_rgba()
    assert_planes(got.planes(), orc.deconstruct_u8(px))


def test_separate_combine_alias_planes(kc, orc):
    h, w = 8, 12
    a = synthetic_rgba(SEED_A, h, w)
    parts = kc.separate_rgba_process(kc.SlotImage.from_planes(a))
    for c in range(4):
        assert_planes(parts[c].planes(), [a[c]])
    comb = kc.combine_rgba_process([parts[3], None, parts[0], None])
    assert_planes(comb.planes(), [a[3], np.zeros_like(a[0]), a[0], np.ones_like(a[0])])
    dflt = kc.separate_rgba_process(None)
    assert [d.size() for d in dflt] == [(1, 1)] * 4


def test_calculate_size_policies(kc, orc):
    sizes = [(128, 128), (256, 64), (64, 256), (128, 128)]
    P = kc.ResizePolicy
    for pol, name in ((P.MostPixels, "MostPixels"), (P.LeastPixels, "LeastPixels"), (P.LargestAxes, "LargestAxes"),
                      (P.SmallestAxes, "SmallestAxes")):
        assert kc.calculate_size(pol, sizes) == orc.calculate_size(name, sizes)
    assert kc.calculate_size(P.MostPixels, []) == (1, 1)
    assert kc.calculate_size(P.SpecificSize((7, 9)), sizes) == (7, 9)


def test_wrapped_torch_memory_roundtrip(kc, orc):
    """Planes can live in caller-owned device memory (torch tensors), zero-copy."""
    import ctypes as C
    import torch
    from kanter_core_amd import _lib
    L = _lib.load()
    h, w = 64, 128
    ta = torch.rand(h, w, device="cuda")
    tb = torch.rand(h, w, device="cuda")
    torch.cuda.synchronize()
    pa, pb = C.c_void_p(), C.c_void_p()
    assert L.kc_plane_wrap(ta.data_ptr(), w, h, w * 4, C.byref(pa)) == 0
    assert L.kc_plane_wrap(tb.data_ptr(), w, h, w * 4, C.byref(pb)) == 0
    ia, ib = C.c_void_p(), C.c_void_p()
    L.kc_image_gray(pa, C.byref(ia))
    L.kc_image_gray(pb, C.byref(ib))
    got = kc.mix_process(kc.SlotImage(ia.value), kc.SlotImage(ib.value), kc.MixType.Multiply).planes()[0]
    L.kc_plane_release(pa)
    L.kc_plane_release(pb)
    assert bit_equal(got, orc.mix_plane("Multiply", ta.cpu().numpy(), tb.cpu().numpy()))
