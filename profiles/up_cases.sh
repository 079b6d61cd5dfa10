# kernel-trace times of the up-sampling / register-tap cases and the fused config #2 kernel on one box
cd /tmp && export TMPDIR=/tmp
for c in "512 4096 Triangle" "512 4096 Lanczos3" "2048 4096 Triangle" "1000 4096 CatmullRom" "4096 2048 Triangle" "4096 3000 Triangle"; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/up_one; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/profiles/resize_one.py $c 20 > $OUT/log 2>&1 || { echo "FAILED $c"; tail -5 $OUT/log; exit 1; }
python3 - "$(find $OUT -name '*kernel_stats.csv' | head -1)" "$c" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "resize" in r["Name"]:
        print("%-24s %-36s avg=%.1f us" % (sys.argv[2], r["Name"][:36], float(r["AverageNs"]) / 1e3))
PY
done
cd $GRAFT_REPO_ROOT && python3 bench.py --workload resize_blend --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('resize_blend kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'], 'parity', d.get('parity'))"
