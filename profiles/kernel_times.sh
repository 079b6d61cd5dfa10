#!/bin/bash
# Kernel-trace durations of the streaming kernels (profiles/stream_kernels.py) under the current build / environment.
#   bash profiles/kernel_times.sh TAG [reps]     (on the GPU box; prints one line per kernel)
set -u
TAG=${1:-run}
REPS=${2:-20}
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/profiles/stream_kernels.py $REPS > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "== $TAG"
python3 - "$f" <<'PY'
import csv, sys
S = 4096 * 4096
alg = {"to_u8_kernel<false": 20, "to_u8_kernel<true": 20, "from_u8_kernel": 20, "height_to_normal_kernel": 16, "fill_kernel": 4}
for r in sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: r["Name"]):
    n = r["Name"]
    if "kc::" not in n and "kc_chain_" not in n:
        continue
    us = float(r["AverageNs"]) / 1e3
    b = next((v for k, v in alg.items() if k in n), None)
    frac = "  %.3f of 8 TB/s" % (b * S / us / 1e6 / 8.0) if b else ""
    print("   %-70s calls=%-4s avg=%7.1f us min=%7.1f us%s" % (n[:70], r["Calls"], us, float(r["MinNs"]) / 1e3, frac))
PY
