cd /tmp && export TMPDIR=/tmp
for mode in 2 0; do for c in "4096 1024 Lanczos3" "4096 2048 Lanczos3" "4096 1024 CatmullRom" "4096 512 Triangle" "3000 700 Gaussian" "4096 3000 Lanczos3"; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/down_one; rm -rf $OUT; mkdir -p $OUT
KC_RESIZE_MODE=$mode timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/profiles/resize_one.py $c 10 4 > $OUT/log 2>&1 || { echo "FAILED $c"; tail -5 $OUT/log; exit 1; }
python3 - "$(find $OUT -name '*kernel_stats.csv' | head -1)" "mode=$mode RGBA | $c" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "resize" in r["Name"]:
        print("%-40s %-30s avg=%.1f us" % (sys.argv[2], r["Name"][:30], float(r["AverageNs"]) / 1e3))
PY
done; done
