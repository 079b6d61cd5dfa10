"""Host side of resize_down2_kernel (csrc/down2.hip; tables built by down2_build in csrc/resize.cpp), no GPU: the dense
per-source-row records and the padded horizontal rows must hold exactly the plain tap table's weights (the same f32 bits, the
same taps, nothing else), and the strip width must keep every strip's source window within 64 column quads.  Reference for
the plain table: image::imageops::resize (crate image 0.24.0) as called from src/shared.rs:159-199."""
import numpy as np
import pytest

import kanter_core_amd as kc

CASES = [(4096, 3000, "Lanczos3"), (4096, 1024, "Lanczos3"), (4096, 1365, "CatmullRom"), (3000, 700, "Gaussian"),
         (4093, 511, "Triangle"), (333, 64, "Lanczos3"), (4096, 2048, "Lanczos3"), (130, 61, "CatmullRom"), (70, 33, "CatmullRom"),
         (64, 16, "Gaussian"), (37, 13, "Gaussian"), (19, 7, "Lanczos3"), (11, 3, "CatmullRom"),
         (4096, 2048, "CatmullRom"), (4096, 2048, "Triangle"), (1030, 515, "Triangle"), (4096, 4000, "Lanczos3"), (300, 290, "Gaussian")]


def plan(i, o, f):
    return kc.resize_down2_plan(i, o, kc.ResizeFilter.parse(f))


@pytest.mark.parametrize("in_n,out_n,filt", CASES)
def test_vertical_records_hold_exactly_the_table(in_n, out_n, filt):
    p = plan(in_n, out_n, filt)
    assert p["stride"] >= 4 and 1 <= p["nc"] <= 4
    left, count, w, rec = p["left"].astype(np.int64), p["count"].astype(np.int64), p["w"], p["vrec"]
    wbits = w.view(np.uint32)
    for g in range(rec.shape[0]):
        rows = range(4 * g, min(out_n, 4 * g + 4))
        lo, hi = min(left[y] for y in rows), max(left[y] + count[y] for y in rows)
        used = int(rec[g, 0, 4])
        assert used == -(-(hi - lo) // 16) and used <= p["nc"]
        assert int(rec[g, 0, 5]) == -(-(hi - lo) // 8)  # the same span in half chunks (the pipelined form's trip count)
        seen = {y: 0 for y in rows}
        for ch in range(p["nc"]):
            r = rec[g, ch]
            s0, last = int(r[0]), int(r[3])
            mask = int(r[1]) | (int(r[2]) << 32)
            assert last == hi - 1 and s0 <= last  # loads are clamped to `last`: never past the group's windows
            if ch >= used:
                assert mask == 0 and not r[8:].any()
                continue
            assert s0 == lo + 16 * ch
            for u in range(16):
                for k in range(4):
                    y, s = 4 * g + k, s0 + u
                    present = y < out_n and left[y] <= s < left[y] + count[y]
                    assert bool(mask >> (4 * u + k) & 1) == present, (g, ch, u, k)
                    if present:
                        assert r[8 + 4 * u + k] == wbits[y, s - left[y]]
                        seen[y] += 1
                    else:
                        assert r[8 + 4 * u + k] == 0  # +0.0: the kernel's fast arm multiplies by it
        assert all(seen[y] == count[y] for y in rows)  # every tap once


@pytest.mark.parametrize("in_n,out_n,filt", CASES)
def test_horizontal_rows_and_strip_width(in_n, out_n, filt):
    p = plan(in_n, out_n, filt)
    if p["hstride"] == 0:
        assert -(-p["stride"] // 4) > 8  # more than 32 taps: the lanes' weight registers do not hold them
        return
    hs, tw = p["hstride"], p["tile_w"]
    assert hs % 4 == 0 and p["stride"] <= hs < p["stride"] + 4
    for x in range(out_n):
        n = int(p["count"][x])
        assert np.array_equal(p["hw"][x, :n].view(np.uint32), p["w"][x, :n].view(np.uint32))
        assert not p["hw"][x, n:].view(np.uint32).any()
    cols_per_lane = 3 if hs // 4 <= 3 else 2 if hs // 4 == 4 else 1
    assert 1 <= tw <= 64 * cols_per_lane
    for x0 in range(0, out_n, tw):
        x1 = min(out_n, x0 + tw)
        c0 = int(p["left"][x0]) & ~3
        quads = (int(p["left"][x1 - 1]) + int(p["count"][x1 - 1]) - c0 + 3) // 4
        assert quads <= 64, (x0, quads)


def test_tables_that_cannot_be_used_say_so():
    assert plan(64, 48, "Nearest")["nc"] == 0           # fewer than 4 taps: the register-tap kernels' ground
    assert plan(64, 48, "Nearest")["hstride"] == 0
    assert plan(48, 64, "Lanczos3")["nc"] == 0          # an up-sampling axis (6 taps)
    assert plan(64, 64, "Lanczos3")["nc"] == 0          # ratio 1
    assert plan(4096, 512, "Gaussian")["nc"] == 0       # 48 taps: four rows' windows span more than 64 samples
    assert plan(333, 41, "Lanczos3")["hstride"] == 0    # 50 taps: more than 32
