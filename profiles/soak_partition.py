"""Host-only soak of the partitioner: random DAGs, every world size and policy, the plan properties of
tests/test_multi_gpu_gloo.py::check_plan.      python profiles/soak_partition.py [graphs]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanter_core_amd as kc
import test_multi_gpu_gloo as mg
from golden_graphs import G

n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(0x50AC0003)
bad = 0
t0 = time.time()
OPS = ["Add", "Subtract", "Multiply", "Divide", "Pow"]
for gi in range(n_graphs):
    g = G()
    outs = []  # (node, slot)
    for e in range(int(rng.integers(1, 5))):
        outs.append((g.add({"Embed": e}), 0))
    for _ in range(int(rng.integers(0, 3))):
        outs.append((g.add({"Value": float(rng.random())}), 0))
    for _ in range(int(rng.integers(1, 40))):
        k = rng.integers(6)
        pick = lambda: outs[rng.integers(len(outs))]
        if k <= 2:
            n = g.add({"Mix": OPS[rng.integers(len(OPS))]})
            for s in range(2):
                if rng.random() < 0.9:
                    o = pick(); g.connect(o[0], n, o[1], s)
            outs.append((n, 0))
        elif k == 3:
            n = g.add("SeparateRgba"); o = pick(); g.connect(o[0], n, o[1], 0)
            outs += [(n, s) for s in range(4)]
        elif k == 4:
            n = g.add("CombineRgba")
            for s in range(4):
                if rng.random() < 0.7:
                    o = pick(); g.connect(o[0], n, o[1], s)
            outs.append((n, 0))
        else:
            n = g.add("HeightToNormal"); o = pick(); g.connect(o[0], n, o[1], 0)
            outs.append((n, 0))
    graph = g.dict()
    root = outs[-1][0]
    lg = mg.host_live_graph(graph)
    for world in (1, 2, 3, 5, 8):
        for policy in (kc.PartitionPolicy.Spread, kc.PartitionPolicy.Auto):
            try:
                plan = lg.partition(root, world, policy)
                mg.check_plan(graph, root, plan, world)
            except Exception as e:  # noqa: BLE001
                bad += 1
                print("FAIL graph %d world %d policy %s: %r" % (gi, world, policy, e), flush=True)
                if bad < 3:
                    print(json.dumps(graph)[:1500])
    if gi % 500 == 499:
        print("%d graphs, %d failures, %.0f s" % (gi + 1, bad, time.time() - t0), flush=True)
print("partition soak finished: %d graphs x 5 worlds x 2 policies, %d failures, %.0f s" % (n_graphs, bad, time.time() - t0))
sys.exit(1 if bad else 0)
