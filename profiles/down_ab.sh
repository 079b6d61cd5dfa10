#!/bin/bash
# Down-sampling cases under KC_DOWN2 = 0 (round 2's kernels: resize_down_kernel / resize_poly_kernel), 1 (resize_down2_kernel
# except where resize_poly_kernel runs at ratio 4 or 8), 2 (resize_down2_kernel everywhere): kernel-trace averages.
#   gpurun -- 'bash profiles/down_ab.sh [planes]'          MODES="2" restricts the modes; other environment
# (KC_DOWN2_VARIANT=1 ...) is passed through to the library
# (The same script measured round 3's rejected LDS-source kernel under KC_DOWN_LDS, r03_down_lds_experiment.txt.)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/down_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "4096 3000 Lanczos3" "4096 1365 CatmullRom" "3000 700 Gaussian" "4096 1024 Lanczos3" "4096 2048 Lanczos3" "4096 2048 CatmullRom" "4096 2048 Triangle" "4096 4000 Lanczos3" "4096 2731 Triangle" "4096 2048 Gaussian" "4096 1024 CatmullRom" "4096 512 Triangle" "4096 512 Gaussian"; do
  for m in ${MODES:-0 1 2}; do
    tag=$(echo $c | tr ' ' '_')_$m
    KC_DOWN2=$m timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $R/profiles/resize_one.py $c 30 ${1:-1} > $OUT/$tag.log 2>&1
    f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$c" "$m" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "resize" in r["Name"]:
        print("%-24s KC_DOWN2=%s %-44s avg=%6.1f us min=%6.1f us" % (sys.argv[2], sys.argv[3], r["Name"].replace("void kc::", "")[:44], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
    rm -rf $OUT/$tag
  done
done
