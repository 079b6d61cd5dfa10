"""Host-side check of the integer-ratio up-sampling plan (csrc/resize.cpp up_axis_build, upsample.h), no GPU:
the class rows the kernels read, pushed through a numpy model of upsample_chain_tile's arithmetic (fixed-length
windows from the unclamped start, zero samples and zero weights outside the source), must reproduce the oracle's
image::imageops::resize bit for bit -- including non-finite samples next to zero weights."""
import numpy as np
import pytest

from util import SEED_A, bit_equal, max_ulp, splitmix_plane

FILTERS = ["Nearest", "Triangle", "CatmullRom", "Gaussian", "Lanczos3"]


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    return kc


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    return orc


def up_class(plan, n_out, o):
    if o < plan["b_lo"]:
        return plan["ratio"] + o
    if o >= n_out - plan["b_hi"]:
        return plan["ratio"] + plan["b_lo"] + (o - (n_out - plan["b_hi"]))
    return o % plan["ratio"]


def model_axis(x, n_out, plan):
    """x: (n_in, m) f32, resampled along axis 0 exactly as the kernel does it (sequential f32 sums from +0.0)."""
    n_in = x.shape[0]
    out = np.zeros((n_out, x.shape[1]), np.float32)
    zero = np.zeros(x.shape[1], np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        for o in range(n_out):
            u = o // plan["ratio"] - plan["off"]
            w = plan["rows"][up_class(plan, n_out, o)]
            acc = np.zeros(x.shape[1], np.float32)
            for j in range(plan["taps"]):
                s = x[u + j] if 0 <= u + j < n_in else zero
                acc = acc + s * w[j]
            out[o] = acc
    return out


def model_resize(p, dw, dh, ph, pv):
    tmp = model_axis(p, dh, pv)                    # vertical pass first, unclamped
    t = model_axis(np.ascontiguousarray(tmp.T), dw, ph).T
    with np.errstate(invalid="ignore"):
        return np.where(t < 0, np.float32(0), np.where(t > 1, np.float32(1), t)).astype(np.float32)  # NaN passes through


@pytest.mark.parametrize("filt", FILTERS)
@pytest.mark.parametrize("src,dst", [((16, 12), (128, 96)), ((5, 7), (20, 7)), ((3, 3), (48, 6)), ((40, 9), (160, 27)),
                                     ((7, 2), (84, 16)), ((1, 5), (8, 40)), ((33, 4), (132, 64))])
def test_plan_reproduces_the_oracle(kc, orc, filt, src, dst):
    (sw, sh), (dw, dh) = src, dst
    f = kc.ResizeFilter.parse(filt)
    ph, pv = kc.resize_upsample_plan(sw, dw, f), kc.resize_upsample_plan(sh, dh, f)
    if ph is None or pv is None:
        # an axis the check rejects (windows longer than the source give an even tap count ...): general kernels
        assert min(sw, sh) <= 6
        return
    assert ph["ratio"] == dw // sw and pv["ratio"] == dh // sh and ph["taps"] % 2 == 1
    p = splitmix_plane(SEED_A, 1, sh, sw) * np.float32(1.5) - np.float32(0.25)
    flat = p.reshape(-1)
    flat[:4] = [np.nan, np.inf, -np.inf, -0.0][:min(4, flat.size)] if flat.size >= 4 else flat[:4]
    p[-1, -1] = np.inf
    got = model_resize(p, dw, dh, ph, pv)
    want = orc.resize_plane(p, dw, dh, filt)
    assert bit_equal(got, want), "%s %s->%s max ulp %s" % (filt, src, dst, max_ulp(got, want))


def test_triangle_8x_has_the_expected_shape(kc):
    p = kc.resize_upsample_plan(512, 4096, kc.ResizeFilter.Triangle)
    assert (p["ratio"], p["taps"], p["off"]) == (8, 3, 1)
    # half of the first and of the last period: windows cut at the border AND a non-zero weight lost, so renormalised
    # (the other half loses a zero weight only and keeps its phase's row)
    assert p["b_lo"] == 4 and p["b_hi"] == 4
    # dyadic phase weights, one of the three exactly zero
    assert p["rows"][0].tolist() == [0.4375, 0.5625, 0.0] and p["rows"][7].tolist() == [0.0, 0.5625, 0.4375]
    assert p["rows"][8].tolist() == [0.0, 1.0, 0.0]  # output 0: the window [-1, 2) cut to [0, 2), weights (1, 0)


@pytest.mark.parametrize("in_n,out_n", [(512, 4095), (512, 768), (4096, 512), (2, 16), (100, 100 * 3 + 1)])
def test_other_tables_are_rejected(kc, in_n, out_n):
    assert kc.resize_upsample_plan(in_n, out_n, kc.ResizeFilter.Triangle) is None


def test_every_filter_and_ratio_plans_or_rejects_consistently(kc):
    for filt in FILTERS:
        f = kc.ResizeFilter.parse(filt)
        for in_n in (1, 2, 3, 5, 8, 31, 64):
            for r in (1, 2, 3, 4, 8, 12, 16, 64):
                p = kc.resize_upsample_plan(in_n, in_n * r, f)
                if p is None:
                    continue
                assert p["ratio"] == r and p["rows"].shape == (r + p["b_lo"] + p["b_hi"], p["taps"])
                assert p["b_lo"] + p["b_hi"] <= in_n * r
                assert np.isfinite(p["rows"]).all()
