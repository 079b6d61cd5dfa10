"""Host cost of building and evaluating very long chains (64 x 64 pixels: the kernels are negligible).
    python profiles/big_graph_host.py [nodes ...]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
import kanter_core_amd as kc
from util import splitmix_plane
import bench
kc.init(0)
for n in [int(a) for a in sys.argv[1:]] or [2000, 20000]:
    for use_cache in (False, True):
        tp = kc.TextureProcessor.new()
        lg = tp.new_live_graph()
        lg.use_cache = use_cache
        a = kc.SlotImage.from_planes([splitmix_plane(1, c, 64, 64) for c in range(4)]); b = kc.SlotImage.from_planes([splitmix_plane(2, c, 64, 64) for c in range(4)])
        t0 = time.perf_counter()
        na = bench.embed(kc, lg, a, 0); nb = bench.embed(kc, lg, b, 1)
        first, last = bench.add_chain(kc, lg, na, nb, n)
        t1 = time.perf_counter()
        lg.await_clean(last); kc.sync()
        t2 = time.perf_counter()
        lg.connect(na, first, 0, 0); lg.await_clean(last); kc.sync()
        t3 = time.perf_counter()
        print("%6d nodes use_cache=%d: build %.3f s, first evaluation %.3f s, re-evaluation %.3f s (%.2f us per node)" % (n, use_cache, t1 - t0, t2 - t1, t3 - t2, (t3 - t2) / n * 1e6))
kc.shutdown()
