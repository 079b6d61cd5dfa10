cd /tmp && export TMPDIR=/tmp
for t in "32 4" "64 4" "64 8" "128 4"; do set -- $t; for c in "4096 512 Triangle" "4096 1024 Lanczos3" "4096 1024 Triangle" "4096 1365 CatmullRom" "3000 700 Gaussian"; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/one_$1_$2; rm -rf $OUT; mkdir -p $OUT
KC_RESIZE_TILE_W=$1 KC_RESIZE_TILE_H=$2 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/profiles/resize_one.py $c 20 > $OUT/log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$t | $c" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "resize" in r["Name"]:
        print("%-40s %-30s avg=%.1f us" % (sys.argv[2], r["Name"][:30], float(r["AverageNs"]) / 1e3))
PY
done; done
