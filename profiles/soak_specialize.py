"""Soak of the run-time specialiser: random chain programs (length 1-60, planes, broadcast constants, operand on either side)
run by the interpreter and by the kernel hiprtc compiles for them; both must equal the oracle bit for bit.
    python profiles/soak_specialize.py [trials]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanter_core_amd as kc
from oracle import oracle as orc
from util import SEED_A, assert_planes, splitmix_plane
import test_gpu_specialize as ts

kc.init(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0x50AC0006)
ops = ["Add", "Subtract", "Multiply", "Divide"]
bad = 0
t0 = time.time()
for trial in range(trials):
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 150))
    planes = [splitmix_plane(SEED_A + i, trial & 3, h, w) * np.float32(0.75) + np.float32(0.125) for i in range(4)]
    imgs = [kc.SlotImage.from_planes([p]) for p in planes]
    n = int(rng.integers(1, 60))
    steps = [(ops[int(rng.integers(0, 4))], int(rng.integers(0, 6)), bool(rng.integers(0, 2))) for _ in range(n)]
    consts = [np.float32(rng.uniform(0.25, 1.0)) for _ in range(n)]

    def build():
        x = imgs[0]
        for (op, k, left), c in zip(steps, consts):
            o = imgs[k] if k < 4 else kc.resize_image(kc.value_process(float(c)), (w, h))
            x = kc.mix_process(o, x, kc.MixType.parse(op)) if left else kc.mix_process(x, o, kc.MixType.parse(op))
        return x

    want = planes[0]
    for (op, k, left), c in zip(steps, consts):
        o = planes[k] if k < 4 else np.full((h, w), c, np.float32)
        want = orc.mix_plane(op, o, want) if left else orc.mix_plane(op, want, o)
    try:
        interp, spec, launches = ts.both_modes(kc, build)
        assert_planes(spec, interp, what="trial %d spec vs interp" % trial)
        assert_planes(spec, [want], what="trial %d vs oracle" % trial)
    except AssertionError as e:
        bad += 1
        print("MISMATCH", str(e)[:200], "n", n, "shape", (h, w), flush=True)
    if trial % 50 == 49:
        print("%d trials, %d mismatches, %.0f s, %s" % (trial + 1, bad, time.time() - t0, kc.specialize_stats()), flush=True)
print("specialiser soak finished: %d trials, %d mismatches, %.0f s" % (trials, bad, time.time() - t0))
kc.shutdown()
sys.exit(1 if bad else 0)
