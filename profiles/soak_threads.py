"""Concurrency soak: several host threads build, evaluate and drop random graphs through the C ABI at once (ctypes releases
the GIL inside every call) and compare with the oracle.  Reference counts and the object free lists are plain data guarded
by the context lock: an entry point that touched them without it would show up here as a crash or a mismatch.
    python profiles/soak_threads.py [threads] [graphs per thread]"""
import os, sys, threading, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import faulthandler
faulthandler.enable()
import numpy as np
import kanter_core_amd as kc
from oracle import oracle as orc
from util import assert_planes
import test_gpu_fuzz_graphs as fz

kc.init(0)
orc.set_threads(1)
n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
per = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = [0] * n_threads
lock = threading.Lock()

def run(t):
    for i in range(per):
        s = 0xF0BB0000 + t * 100000 + i
        try:
            _, _, requested = fz._build(kc, orc, s)
            for n in requested[:2]:
                lg, ref, _ = fz._build(kc, orc, s)
                try:
                    want = ref.node_slot_datas(int(n))
                except (RuntimeError, AssertionError):
                    try:
                        lg.await_clean(n)
                        bad[t] += 1
                    except kc.TexProError:
                        pass
                    continue
                got = lg.await_clean(n).node_slot_datas(n)
                assert len(got) == len(want)
                for g, w in zip(sorted(got, key=lambda x: x.slot_id), sorted(want, key=lambda x: x.slot_id)):
                    assert_planes(g.image.planes(), w.image.planes, what="thread %d seed %x" % (t, s))
                # plain operator calls and explicit releases from this thread as well
                img = got[0].image
                kc.resize_image(img, (7, 5)).planes()
                img.to_u8()
        except AssertionError as e:
            bad[t] += 1
            with lock:
                print("MISMATCH", str(e)[:200], flush=True)

t0 = time.time()
ths = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
for th in ths: th.start()
for th in ths: th.join()
print("thread soak: %d threads x %d graphs, %d failures, %.0f s" % (n_threads, per, sum(bad), time.time() - t0))
kc.shutdown()
sys.exit(1 if sum(bad) else 0)
