# kernel-trace times of to_u8 (plain / sRGB) on four resident 4096^2 planes:  gpurun -- 'bash profiles/to_u8_times.sh'
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/to_u8; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import kanter_core_amd as kc
from util import SEED_A, splitmix_plane
kc.init(0)
img = kc.SlotImage.from_planes([splitmix_plane(SEED_A, c, 4096, 4096) for c in range(4)])
for _ in range(10):
    img.to_u8(False); img.to_u8(True)
kc.sync()
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $OUT/run.py > $OUT/log 2>&1
python3 - "$(find $OUT -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "to_u8" in r["Name"]:
        print("%-60s calls=%s avg=%.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
