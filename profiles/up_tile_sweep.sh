#!/bin/bash
# resize_lds_kernel on non-integer up-sampling: tile shape (KC_RESIZE_TILE_W x KC_RESIZE_TILE_H) against time.
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/up_tiles; mkdir -p $OUT
for c in "3000 4096 Lanczos3" "1000 4096 CatmullRom" "2048 4096 CatmullRom" "2048 4096 Triangle" "700 3000 Gaussian"; do
  for tw in 1024 512 256 128 64; do
    for th in 8 16 32; do
      tag=$(echo "$c $tw $th" | tr ' ' '_')
      KC_RESIZE_TILE_W=$tw KC_RESIZE_TILE_H=$th timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $R/profiles/resize_one.py $c 20 1 > $OUT/$tag.log 2>&1
      f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
      python3 - "$f" "$c" "$tw" "$th" <<'PY'
import csv, sys
src, dst = [int(v) for v in sys.argv[2].split()[:2]]
for r in csv.DictReader(open(sys.argv[1])):
    if "resize" in r["Name"] or "upsample" in r["Name"]:
        us = float(r["AverageNs"]) / 1e3
        print("%-22s tile %4s x %2s  %-30s avg=%6.1f us  %.2f" % (sys.argv[2], sys.argv[3], sys.argv[4], r["Name"].replace("void kc::", "")[:30], us, 4.0 * (src * src + dst * dst) / us / 1e6 / 8.0))
PY
      rm -rf $OUT/$tag
    done
  done
done
