set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/poly_rows
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "4096 512 Gaussian" "2048 512 Lanczos3" "4096 1024 Lanczos3" "2048 256 Gaussian" "1024 256 Lanczos3"; do
 for r in 4 8 12 16; do
  tag=$(echo "$c $r" | tr ' ' '_')
  KC_STREAM_DOWN=0 KC_POLY_ROWS=$r timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $R/profiles/resize_one.py $c 30 1 > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$c" "$r" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "resize" in r["Name"]:
        print("%-22s rows=%-3s %-40s avg=%6.1f us min=%6.1f us" % (sys.argv[2], sys.argv[3], r["Name"].replace("void kc::", "")[:40], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  rm -rf $OUT/$tag
 done
done
