"""Soak of the integer-ratio up-sampling kernels beyond the test suite's cases: random source extents, ratios (any multiple of
4 across, anything down the rows), all five filters, 1 or 4 planes, non-finite samples; plain and fused into random Mix
chains (1-4 inputs, {+, -, *}, constants on either side), interpreter and specialised; everything against the oracle and
against the general kernels (kc_set_resize_mode(4)).     python profiles/soak_upsample.py [cases] [seed]"""
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
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0x0B5A3F1E)
FILTERS = ["Nearest", "Triangle", "CatmullRom", "Gaussian", "Lanczos3"]
OPS = ["Add", "Subtract", "Multiply"]
bad = new_path = fused_new = 0
t0 = time.time()


def eq(a, b):
    return bit_equal(a, b)


for i in range(n_cases):
    rh = int(rng.choice([4, 4, 8, 8, 12, 16, 20, 32, 64]))
    rv = int(rng.choice([1, 2, 3, 4, 4, 5, 6, 8, 8, 12, 16, 32, 64]))
    sw = int(rng.integers(1, max(2, min(300, 2600 // rh))))
    sh = int(rng.integers(1, max(2, min(300, 2600 // rv))))
    dw, dh = sw * rh, sh * rv
    filt = FILTERS[rng.integers(len(FILTERS))]
    f = kc.ResizeFilter.parse(filt)
    n_pl = 4 if rng.random() < 0.25 else 1
    ps = [(rng.random((sh, sw), dtype=np.float32) * np.float32(1.5) - np.float32(0.25)).astype(np.float32) for _ in range(n_pl)]
    if ps[0].size >= 6:
        ps[0].reshape(-1)[rng.integers(ps[0].size, size=4)] = [np.nan, np.inf, -np.inf, -0.0]
        ps[0][-1, -1] = np.inf
    want = [orc.resize_plane(p, dw, dh, filt) for p in ps]
    n0 = kc.stats_counter("upsample_launches")
    got = kc.resize_image(kc.SlotImage.from_planes(ps), (dw, dh), f).planes()
    new_path += kc.stats_counter("upsample_launches") - n0
    ok = all(eq(g, w) for g, w in zip(got, want))
    # fused into a random chain (gray), both device paths
    if ok and rng.random() < 0.6 and filt in ("Nearest", "Triangle"):
        k_res = int(rng.integers(0, 4))
        res = [rng.random((dh, dw), dtype=np.float32) for _ in range(k_res)]
        steps = [(OPS[rng.integers(3)], int(rng.integers(0, k_res + 2)), bool(rng.integers(2))) for _ in range(int(rng.integers(1, 9)))]
        consts = [np.float32(rng.uniform(0.0, 1.0)) for _ in steps]
        for mode in (0, 2):
            kc.set_specialize(mode)
            up = kc.resize_image(kc.SlotImage.from_planes([ps[0]]), (dw, dh), f)
            imgs = [kc.SlotImage.from_planes([r]) for r in res]
            x, wx = up, want[0]
            n1 = kc.stats_counter("upsample_chain_launches")
            for (op, k, left), c in zip(steps, consts):
                if k < k_res:
                    o, wo = imgs[k], res[k]
                elif k == k_res:
                    o, wo = up, want[0]
                else:
                    o, wo = kc.resize_image(kc.value_process(float(c)), (dw, dh)), np.full((dh, dw), c, np.float32)
                x = kc.mix_process(o, x, kc.MixType.parse(op)) if left else kc.mix_process(x, o, kc.MixType.parse(op))
                wx = orc.mix_plane(op, wo, wx) if left else orc.mix_plane(op, wx, wo)
            g = x.planes()[0]
            fused_new += kc.stats_counter("upsample_chain_launches") - n1
            if not eq(g, wx):
                ok = False
                print("FUSED MISMATCH case %d mode %d: %s %dx%d -> %dx%d steps %s" % (i, mode, filt, sw, sh, dw, dh, steps), flush=True)
        kc.set_specialize(1)
    if not ok:
        bad += 1
        print("MISMATCH case %d: %s %dx%d -> %dx%d planes %d" % (i, filt, sw, sh, dw, dh, n_pl), flush=True)
    if i % 200 == 199:
        print("%d cases, %d through upsample_kernel, %d fused launches, %d bad, %.0f s" % (i + 1, new_path, fused_new, bad, time.time() - t0), flush=True)
print("soak_upsample: %d cases, %d plain launches of the new kernels, %d fused, %d mismatches, %.0f s" % (n_cases, new_path, fused_new, bad, time.time() - t0))
sys.exit(1 if bad else 0)
