#!/bin/bash
# Up-sampling and slight down-sampling through the register-tap kernels (resize_lds_kernel / upsample_kernel), kernel-trace averages,
# one plane unless $1 says 4.   gpurun -- 'bash profiles/up_cases.sh [planes]'     (extra environment is passed through)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/up_cases
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "1000 4096 CatmullRom" "1000 4096 Lanczos3" "1000 4096 Triangle" "2048 4096 Triangle" "2048 4096 CatmullRom" "3000 4096 Lanczos3" "3000 4096 Gaussian" "4096 4000 Lanczos3" "4096 3500 CatmullRom" "512 4096 Lanczos3" "1365 4096 Triangle" "700 3000 Gaussian"; do
   tag=$(echo "$c" | tr ' ' '_')
   timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $R/profiles/resize_one.py $c 30 ${1:-1} > $OUT/$tag.log 2>&1
   f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
   python3 - "$f" "$c" "${1:-1}" <<'PY'
import csv, sys
src, dst = [int(v) for v in sys.argv[2].split()[:2]]
for r in csv.DictReader(open(sys.argv[1])):
    if "resize" in r["Name"] or "upsample" in r["Name"]:
        us = float(r["AverageNs"]) / 1e3
        print("%-22s %-44s avg=%6.1f us min=%6.1f us  %.2f of 8 TB/s" % (sys.argv[2], r["Name"].replace("void kc::", "")[:44], us, float(r["MinNs"]) / 1e3, int(sys.argv[3]) * 4.0 * (src * src + dst * dst) / us / 1e6 / 8.0))
PY
   rm -rf $OUT/$tag
done
