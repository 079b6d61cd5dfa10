#!/bin/bash
# The integer-ratio down-sampling cases of resize_poly_kernel (and the ratio-2 cases resize_down2_kernel takes), kernel-trace averages,
# one plane unless $1 says 4.   gpurun -- 'bash profiles/poly_cases.sh [planes]'     (extra environment is passed through)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/poly_cases
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "4096 512 Gaussian" "4096 512 CatmullRom" "4096 512 Triangle" "4096 1024 Lanczos3" "4096 1024 CatmullRom" "4096 2048 Lanczos3" "4096 2048 Gaussian" "2048 512 Lanczos3" "2048 256 Gaussian" "1024 256 Lanczos3" "8192 1024 Gaussian"; do
   tag=$(echo "$c" | tr ' ' '_')
   timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $R/profiles/resize_one.py $c 30 ${1:-1} > $OUT/$tag.log 2>&1
   f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
   python3 - "$f" "$c" "${1:-1}" <<'PY'
import csv, sys
src, dst = [int(v) for v in sys.argv[2].split()[:2]]
for r in csv.DictReader(open(sys.argv[1])):
    if "resize" in r["Name"]:
        us = float(r["AverageNs"]) / 1e3
        print("%-22s %-40s avg=%6.1f us min=%6.1f us  %.2f of 8 TB/s" % (sys.argv[2], r["Name"].replace("void kc::", "")[:40], us, float(r["MinNs"]) / 1e3, int(sys.argv[3]) * 4.0 * (src * src + dst * dst) / us / 1e6 / 8.0))
PY
   rm -rf $OUT/$tag
done
