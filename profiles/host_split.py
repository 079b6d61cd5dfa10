#!/usr/bin/env python3
"""Host time of the two calls of one re-evaluation of the 32-node graph (connect + await_clean), separately, at a size where
the kernel is short: python profiles/host_split.py [size] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kanter_core_amd as kc
from bench import add_chain, embed
from util import SEED_A, SEED_B, splitmix_plane

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
kc.init(0)
tp = kc.TextureProcessor.new()
lg = tp.new_live_graph()
ia = kc.SlotImage.from_planes([splitmix_plane(SEED_A, c, S, S) for c in range(4)])
ib = kc.SlotImage.from_planes([splitmix_plane(SEED_B, c, S, S) for c in range(4)])
na, nb = embed(kc, lg, ia, 0), embed(kc, lg, ib, 1)
first, last = add_chain(kc, lg, na, nb, 32)
for _ in range(50):
    lg.connect(na, first, 0, 0)
    lg.await_clean(last)
kc.specialize_wait()
for _ in range(200):
    lg.connect(na, first, 0, 0)
    lg.await_clean(last)
kc.sync()
tc = ta = 0.0
t_all0 = time.perf_counter()
for i in range(reps):
    t0 = time.perf_counter()
    lg.connect(na, first, 0, 0)
    t1 = time.perf_counter()
    lg.await_clean(last)
    t2 = time.perf_counter()
    tc += t1 - t0
    ta += t2 - t1
    if i % 64 == 63:
        kc.sync()  # keep the queue short: host time only
t_all = time.perf_counter() - t_all0
print("size %d: connect %.2f us, await_clean %.2f us, loop %.2f us per evaluation; replayed %d of %d; option replay=%d" % (
    S, tc / reps * 1e6, ta / reps * 1e6, t_all / reps * 1e6, kc.stats_counter("replayed_evaluations"), reps + 250, kc.get_option("replay")))
