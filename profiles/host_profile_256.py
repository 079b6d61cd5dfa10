import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import time
import kanter_core_amd as kc
from util import splitmix_plane
import bench
kc.init(0)
S=256
tp=kc.TextureProcessor.new()
a=kc.SlotImage.from_planes([splitmix_plane(1,c,S,S) for c in range(4)]); b=kc.SlotImage.from_planes([splitmix_plane(2,c,S,S) for c in range(4)])
lg=tp.new_live_graph()
na=bench.embed(kc,lg,a,0); nb=bench.embed(kc,lg,b,1)
first,last=bench.add_chain(kc,lg,na,nb,32)
for _ in range(50):
    lg.connect(na,first,0,0); lg.await_clean(last)
kc.specialize_wait()
n=int(os.environ.get('KC_EVALS','5000'))
t0=time.perf_counter()
for _ in range(n):
    lg.connect(na,first,0,0); lg.await_clean(last)
kc.sync()
print("host us per evaluation: %.2f" % ((time.perf_counter()-t0)/n*1e6))
kc.shutdown()
