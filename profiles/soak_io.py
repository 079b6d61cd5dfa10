"""Soak of the u8 boundary and the PNG codec: random shapes / channel counts through from_u8 and to_u8 (plain and sRGB) against
the oracle; random 8-bit gray / gray+alpha / RGB / RGBA PNGs (plain and Adam7, every filter type) written by the test writer,
read by the library; the library's own writer read back.      python profiles/soak_io.py [cases]"""
import os, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanter_core_amd as kc
from oracle import oracle as orc
import test_gpu_png_variants as pv

kc.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(0x50AC0005)
bad = 0
t0 = time.time()
tmp = tempfile.mkdtemp()
for i in range(n):
    h, w, ch = int(rng.integers(1, 90)), int(rng.integers(1, 90)), int(rng.integers(1, 5))
    px = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    got = kc.SlotImage.from_u8(px).planes()
    want = orc.deconstruct_u8(px)
    if not all(np.array_equal(a, b) for a, b in zip(got, want)):
        bad += 1; print("MISMATCH from_u8", h, w, ch, flush=True)
    # to_u8 on float planes with out-of-range and special values
    planes = [(rng.random((h, w), dtype=np.float32) * np.float32(1.4) - np.float32(0.2)).astype(np.float32) for _ in range(4 if ch > 1 else 1)]
    planes[0].reshape(-1)[rng.integers(h * w, size=min(4, h * w))] = [np.nan, np.inf, -np.inf, -0.0][:min(4, h * w)]
    for srgb in (False, True):
        g = kc.SlotImage.from_planes(planes).to_u8(srgb)
        wnt = orc.to_u8(orc.Image(planes), srgb)
        if not np.array_equal(g, wnt):
            bad += 1; print("MISMATCH to_u8 srgb=%s" % srgb, h, w, len(planes), flush=True)
    # PNG: the test writer's file read by the library
    color, chans = [(0, 1), (4, 2), (2, 3), (6, 4)][ch - 1]
    path = os.path.join(tmp, "a.png")
    pv.write_png(path, px, color, 8, interlace=bool(rng.integers(2)))
    gotp = kc.SlotImage.read_png(path).planes()
    u8 = [px[:, :, c] for c in range(chans)]  # deconstruct_image: channel c -> plane c (a gray file fills R only)
    wantp = pv.expect(u8, h, w)
    if not all(np.array_equal(a, b) for a, b in zip(gotp, wantp)):
        bad += 1; print("MISMATCH read_png color", color, h, w, flush=True)
    # the library's writer, read back by the library
    img = kc.SlotImage.from_u8(px if ch == 4 else np.concatenate([px, np.full((h, w, 4 - ch), 255, np.uint8)], axis=2))
    p2 = os.path.join(tmp, "b.png")
    img.write_png(p2)
    back = kc.SlotImage.read_png(p2)
    if not np.array_equal(back.to_u8(), img.to_u8()):
        bad += 1; print("MISMATCH write/read png", h, w, flush=True)
    if i % 200 == 199:
        print("%d cases, %d mismatches, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
print("io soak finished: %d cases, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
kc.shutdown()
sys.exit(1 if bad else 0)
