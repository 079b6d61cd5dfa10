"""Soak of resize_down2_kernel (csrc/down2.hip) beyond the test suite's cases: random source extents and down-sampling ratios
(1.01 .. 9 per axis, independently; integer ratios among them), the four filters with windows of 4 taps or more, 1 or 4 planes,
sources with infinities / NaNs / values that overflow in the vertical sums / denormals; every result against the CPU oracle
and against the kernels it replaces (kc_set_option("down2", 0)); whole images and row bands.
    python profiles/soak_down2.py [cases] [seed]"""
import faulthandler, os, sys, time
faulthandler.enable()
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanter_core_amd as kc
from oracle import oracle as orc
from util import bit_equal

kc.init(0)
orc.set_threads(8)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0xD0D02)
FILTERS = ["Triangle", "CatmullRom", "Gaussian", "Lanczos3"]
bad = reached = bands = 0
t0 = time.time()


def extent(r):
    """(source, destination) extents for one axis at about ratio r."""
    d = int(rng.integers(1, 900))
    if rng.random() < 0.3:
        s = d * int(rng.choice([2, 2, 3, 4, 4, 5, 8]))  # an integer ratio
    else:
        s = max(d + 1, int(round(d * r)))
    return min(s, 3000), d


for i in range(n_cases):
    filt = FILTERS[rng.integers(len(FILTERS))]
    f = kc.ResizeFilter.parse(filt)
    sw, dw = extent(rng.uniform(1.01, 9.0))
    sh, dh = extent(rng.uniform(1.01, 9.0))
    n_pl = 4 if rng.random() < 0.2 else 1
    ps = [(rng.random((sh, sw), dtype=np.float32) * np.float32(1.5) - np.float32(0.25)).astype(np.float32) for _ in range(n_pl)]
    kind = rng.integers(4)
    p0 = ps[0]
    if kind == 1 and p0.size >= 8:
        p0.reshape(-1)[rng.integers(p0.size, size=4)] = [np.nan, np.inf, -np.inf, -0.0]
        p0[-1, -1] = np.inf
    elif kind == 2 and p0.size >= 8:
        y, x = int(rng.integers(sh)), int(rng.integers(sw))
        p0[max(0, y - 3):y + 3, max(0, x - 2):x + 2] = np.float32(3.0e38) * (1 if rng.random() < 0.5 else -1)
    elif kind == 3:
        p0[::5, ::3] = np.float32(1e-42)
    want = [orc.resize_plane(p, dw, dh, filt) for p in ps]
    ok = True
    for mode in (1, 2, 0):
        kc.set_option("down2", mode)
        n0 = kc.stats_counter("down2_launches")
        got = kc.resize_image(kc.SlotImage.from_planes(ps), (dw, dh), f).planes()
        if mode == 2:
            reached += kc.stats_counter("down2_launches") - n0
        if not all(bit_equal(g, w) for g, w in zip(got, want)):
            ok = False
            print("MISMATCH case %d down2=%d: %s %dx%d -> %dx%d planes %d kind %d" % (i, mode, filt, sw, sh, dw, dh, n_pl, kind), flush=True)
    kc.set_option("down2", 1)
    # the same resample by row bands through a graph: Mix(Add) of the source with a zero image of the target size under
    # LeastPixels resamples the source (x + 0.0 == x for every x a resample can produce: it is never -0.0)
    if ok and dh >= 8 and rng.random() < 0.25:
        tp = kc.TextureProcessor.new()
        lg = tp.new_live_graph()
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes([ps[0]])), 0)
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes([np.zeros((dh, dw), np.float32)])), 1)
        e0 = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
        e1 = lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
        mix = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)).with_resize_policy(kc.ResizePolicy.LeastPixels).with_resize_filter(f))
        lg.connect(e0, mix, 0, 0)
        lg.connect(e1, mix, 0, 1)
        y0 = int(rng.integers(0, dh - 4))
        y1 = int(rng.integers(y0 + 1, dh + 1))
        band = lg.evaluate_band(mix, y0, y1).planes()[0]
        bands += 1
        if not bit_equal(band, want[0][y0:y1]):
            ok = False
            print("BAND MISMATCH case %d: %s %dx%d -> %dx%d rows %d..%d" % (i, filt, sw, sh, dw, dh, y0, y1), flush=True)
    if not ok:
        bad += 1
    if i % 100 == 99:
        print("%d cases, %d through resize_down2_kernel (down2 = 2), %d bands, %d bad, %.0f s" % (i + 1, reached, bands, bad, time.time() - t0), flush=True)
print("soak_down2: %d cases, %d launches of resize_down2_kernel under down2 = 2, %d bands, %d mismatches, %.0f s" % (n_cases, reached, bands, bad, time.time() - t0))
sys.exit(1 if bad else 0)
