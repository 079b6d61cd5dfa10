#!/bin/bash
# Tile shape of the fused resize + chain kernel on BASELINE config #2 (bench.py --workload resize_blend), kernel us by HIP events.
for t in "0 0" "1024 8" "1024 16" "1024 32" "1024 64" "512 16" "512 32" "512 64" "256 32" "256 64"; do
  set -- $t
  echo -n "tile $1 x $2: "
  KC_RESIZE_TILE_W=$1 KC_RESIZE_TILE_H=$2 python bench.py --workload resize_blend --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel_us'], r['frac'], r['launches_per_step'])"
done
