# A/B of the down-sampling kernels on one box: KC_RESIZE_MODE=2 = resize_wide_kernel, 1 = resize_down_kernel only,
# 0 (default) = resize_poly_kernel where the vertical table is regular.
#   gpurun -- 'bash profiles/down_cases.sh > gpurun_out/down_cases.txt'      (optional: MODES="0" TILE_H="16 32")
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-2 1 0}; do for th in ${TILE_H:-0}; do for c in "4096 1024 Lanczos3" "4096 512 Triangle" "4096 1365 CatmullRom" "3000 700 Gaussian" "4096 2048 Lanczos3" "4096 3000 Lanczos3" "4096 1024 CatmullRom" "4096 512 Gaussian"; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/down_one; rm -rf $OUT; mkdir -p $OUT
KC_RESIZE_MODE=$mode KC_RESIZE_TILE_H=$th timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/profiles/resize_one.py $c 20 > $OUT/log 2>&1 || { echo "FAILED mode=$mode $c"; tail -5 $OUT/log; exit 1; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" "mode=$mode tile_h=$th | $c" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "resize" in r["Name"]:
        print("%-44s %-28s avg=%.1f us" % (sys.argv[2], r["Name"][:28], float(r["AverageNs"]) / 1e3))
PY
done; done; done
