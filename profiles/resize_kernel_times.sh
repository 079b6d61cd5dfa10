#!/bin/bash
# True kernel durations (rocprofv3 kernel trace) of the resize kernels for a few cases; the Python-side
# sweep (resize_tile_sweep.py) includes host call overhead, which matters below ~20 us.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/rk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in ${RESIZE_CASES:-"512 4096 Triangle" "512 4096 Lanczos3" "1024 4096 Triangle" "1024 4096 CatmullRom" "2048 4096 Triangle" "1000 4096 CatmullRom" "4096 512 Triangle" "4096 2048 Triangle" "4096 1024 Lanczos3" "4096 3000 Triangle" "4096 3000 Lanczos3" "4096 1365 CatmullRom" "3000 700 Gaussian"}; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/profiles/resize_one.py $c 30 > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $c"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "resize" in r["Name"] or "upsample" in r["Name"]:
        print("   %-60s calls=%s avg=%.1f us min=%.1f us" % (r["Name"].replace("void kc::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
