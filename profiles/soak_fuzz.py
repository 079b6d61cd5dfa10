"""One-process soak beyond the seeds of tests/test_gpu_fuzz_graphs.py: resizes biased towards integer ratios and larger
extents (the streaming / wave-uniform down-sampling kernels, band and strip edges), and more random graphs.
    python profiles/soak_fuzz.py [resizes] [graphs]"""
import faulthandler, os, sys, time
faulthandler.enable()
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanter_core_amd as kc
from oracle import oracle as orc
from util import bit_equal, assert_planes
import test_gpu_fuzz_graphs as fz

kc.init(0)
orc.set_threads(8)
n_resize = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n_graphs = int(sys.argv[2]) if len(sys.argv) > 2 else 600
graph0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # first graph seed offset
verbose = len(sys.argv) > 4
FILTERS = ["Nearest", "Triangle", "CatmullRom", "Gaussian", "Lanczos3"]
rng = np.random.default_rng(0x50AC0001)
bad = 0
t0 = time.time()
for i in range(n_resize):
    dw, dh = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    kind = rng.integers(4)
    def src_of(d):
        if kind == 0:
            return d * int(rng.choice([2, 3, 4, 8]))                      # exact integer ratio
        if kind == 1:
            return max(1, d * int(rng.choice([2, 4, 8])) + int(rng.integers(-3, 4)))  # just off an integer ratio
        if kind == 2:
            return max(1, int(d * rng.uniform(1.0, 9.0)))                   # any down-sampling ratio
        return max(1, int(d * rng.uniform(0.1, 1.0)))                       # up-sampling
    sw, sh = min(src_of(dw), 3200), min(src_of(dh), 3200)
    filt = FILTERS[rng.integers(len(FILTERS))]
    planes = 4 if rng.random() < 0.25 else 1
    ps = [(rng.random((sh, sw), dtype=np.float32) * np.float32(1.5) - np.float32(0.25)).astype(np.float32) for _ in range(planes)]
    if ps[0].size >= 8:
        ps[0].reshape(-1)[rng.integers(ps[0].size, size=4)] = [np.nan, np.inf, -np.inf, -0.0]
    got = kc.resize_image(kc.SlotImage.from_planes(ps), (dw, dh), kc.ResizeFilter.parse(filt)).planes()
    for c in range(planes):
        want = orc.resize_plane(ps[c], dw, dh, filt)
        if not bit_equal(got[c], want):
            bad += 1
            print("MISMATCH resize %s %dx%d -> %dx%d plane %d of %d" % (filt, sw, sh, dw, dh, c, planes), flush=True)
    if i % 250 == 249:
        print("resizes: %d done, %d mismatches, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
for seed in range(graph0, graph0 + n_graphs):
    s = 0xF0990000 + seed
    if verbose:
        print('seed', seed, flush=True)
    _, _, requested = fz._build(kc, orc, s)
    for n in requested:
        lg, ref, _ = fz._build(kc, orc, s)
        try:
            want = ref.node_slot_datas(int(n))
        except (RuntimeError, AssertionError):
            try:
                lg.await_clean(n)
                bad += 1
                print("graph seed %x node %d: the reference fails, the library does not" % (s, int(n)), flush=True)
            except kc.TexProError:
                pass
            continue
        got = lg.await_clean(n).node_slot_datas(n)
        try:
            assert len(got) == len(want)
            for g, w in zip(sorted(got, key=lambda x: x.slot_id), sorted(want, key=lambda x: x.slot_id)):
                assert int(g.slot_id) == int(w.slot_id) and g.image.is_rgba() == w.image.is_rgba
                assert_planes(g.image.planes(), w.image.planes, what="seed %x node %d" % (s, int(n)))
        except AssertionError as e:
            bad += 1
            print("MISMATCH graph seed %x node %d: %s" % (s, int(n), str(e)[:200]), flush=True)
    if seed % 100 == 99:
        print("graphs: %d done, %d mismatches, %.0f s" % (seed + 1, bad, time.time() - t0), flush=True)
# row bands of random graphs, stacked, against the whole-image evaluation (both by the library; csrc/bands.cpp)
n_band = int(os.environ.get("KC_SOAK_BANDS", "0"))
brng = np.random.default_rng(0x50AC0002)
unsupported = 0
band0 = int(os.environ.get("KC_SOAK_BAND0", "0"))
for seed in range(band0, band0 + n_band):
    sd = 0xF0AA0000 + seed
    if os.environ.get("KC_SOAK_VERBOSE"):
        print("band seed", seed, flush=True)
    _, _, requested = fz._build(kc, orc, sd)
    for n in requested[:2]:
        lg, _, _ = fz._build(kc, orc, sd)
        try:
            whole = lg.await_clean(n).node_slot_datas(n)
        except kc.TexProError:
            continue
        for w in whole:
            planes = w.image.planes()
            h = planes[0].shape[0]
            cuts = sorted(set([0, h] + [int(c) for c in brng.integers(0, h + 1, size=int(brng.integers(0, 4)))]))
            lg2, _, _ = fz._build(kc, orc, sd)
            try:
                parts = [lg2.evaluate_band(n, y0, y1, int(w.slot_id)).planes() for y0, y1 in zip(cuts[:-1], cuts[1:]) if y1 > y0]
            except kc.TexProError as e:
                unsupported += 1
                if os.environ.get("KC_SOAK_VERBOSE"):
                    print("   refused:", str(e)[:160], flush=True)
                continue
            try:
                stacked = [np.concatenate([p[c] for p in parts], axis=0) for c in range(len(planes))]
                assert_planes(stacked, planes, what="bands seed %x node %d slot %d cuts %s" % (sd, int(n), int(w.slot_id), cuts))
            except (AssertionError, ValueError) as e:
                bad += 1
                print("MISMATCH", str(e)[:240], "| whole", [p.shape for p in planes], "parts", [[q.shape for q in p] for p in parts], flush=True)
    if seed % 500 == 499:
        print("bands: %d graphs done, %d mismatches, %d band evaluations refused, %.0f s" % (seed + 1, bad, unsupported, time.time() - t0), flush=True)

# the edit / re-evaluate and the fused-vs-unfused-vs-cached tests of the suite, on seeds beyond the suite's
n_edit = int(os.environ.get("KC_SOAK_EDITS", "0"))
for seed in range(80, 80 + n_edit):
    try:
        fz.test_random_graph_edits_re_evaluate_like_a_fresh_graph(kc, orc, seed)
        if seed < 80 + n_edit // 4:
            fz.test_random_graph_unfused_and_cached_agree(kc, orc, 40 + seed)
    except AssertionError as e:
        bad += 1
        print("MISMATCH edit / cache seed %d: %s" % (seed, str(e)[:200]), flush=True)
    if seed % 500 == 499:
        print("edits: %d done, %d mismatches, %.0f s" % (seed + 1 - 80, bad, time.time() - t0), flush=True)
print("soak finished: %d resizes, %d graphs, %d edit sequences, %d mismatches, %.0f s" % (n_resize, n_graphs, n_edit, bad, time.time() - t0))
kc.shutdown()
sys.exit(1 if bad else 0)
