#!/bin/bash
# HBM traffic (FETCH_SIZE x2 + WRITE_SIZE, KiB) of the resize kernels for a few cases, against the algorithmic bytes.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/rt
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "512 4096 Triangle" "4096 512 Triangle" "4096 2048 Triangle" "4096 1024 Lanczos3" "4096 3000 Triangle"; do
  tag=$(echo $c | tr ' ' '_')
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${tag}_$ctr -- python3 $GRAFT_REPO_ROOT/profiles/resize_one.py $c 12 > $OUT/${tag}_$ctr.log 2>&1
  done
  python3 - "$OUT" "$tag" $c <<'PY'
import csv, glob, sys
out, tag, s, d = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
v = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, tag, ctr), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if "resize" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        rows = [per[k] for k in sorted(per)][2:]
    v[ctr] = sum(rows) / max(len(rows), 1)
fetch, write = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
alg_r, alg_w = 4.0 * s * s, 4.0 * d * d
print("%-22s read %7.1f MB (algorithmic %6.1f)  written %6.1f MB (algorithmic %6.1f)" % (tag, fetch / 1e6, alg_r / 1e6, write / 1e6, alg_w / 1e6))
PY
done
