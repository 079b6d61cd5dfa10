#!/usr/bin/env python3
"""Soak of the native multi-rank path: N seeded random graphs (tests/test_gpu_fuzz_graphs.py's generator) through a branch plan and a
band plan with WORLD processes on one GPU, home result against the oracle.   python profiles/soak_ranks.py [N=400] [WORLD=3] [FIRST_SEED]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    first = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0xF0550000
    import kanter_core_amd as kc
    from oracle import oracle as orc
    from rank_harness import run_ranks
    from test_gpu_fuzz_graphs import _build
    kc.init(0)
    seeds, want = [], {}
    seed = first
    while len(seeds) < n and seed < first + 20 * n:
        _, ref, requested = _build(kc, orc, seed)
        try:
            sds = ref.node_slot_datas(int(requested[0]))
            if sds:
                first_slot = sorted(sds, key=lambda s: s.slot_id)[0]
                want[seed] = [np.ascontiguousarray(p).tobytes() for p in first_slot.image.planes]
                seeds.append(seed)
        except (RuntimeError, AssertionError):
            pass
        seed += 1
    bad = checked = moved = bands = 0
    for lo in range(0, len(seeds), 50):
        chunk = seeds[lo:lo + 50]
        outs = run_ranks(world, "fuzz_plans", timeout=900, seeds=chunk)
        for s in chunk:
            for name in ("spread", "bands"):
                r0 = outs[0][s][name]
                if isinstance(r0, str):
                    continue
                checked += 1
                moved += r0["transfers"] > 0
                bands += name == "bands"
                got = r0["planes"]
                ok = got is not None and len(got) == len(want[s])
                for g, x in zip(got or [], want[s]):
                    if g != x:
                        ga, xa = np.frombuffer(g, np.uint32), np.frombuffer(x, np.uint32)
                        ok &= bool(((ga == xa) | (np.isnan(ga.view(np.float32)) & np.isnan(xa.view(np.float32)))).all())
                if not ok:
                    bad += 1
                    print("MISMATCH seed %#x plan %s kind %s transfers %s levels %s" % (s, name, r0["kind"], r0["transfers"], r0["levels"]), flush=True)
        print("%d / %d seeds done" % (min(lo + 50, len(seeds)), len(seeds)), flush=True)
    print("soak_ranks: %d graphs, world %d: %d plan evaluations checked (%d with transfers, %d band plans), %d mismatches" % (len(seeds), world, checked, moved, bands, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
