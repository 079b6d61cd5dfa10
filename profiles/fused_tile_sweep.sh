#!/bin/bash
# Tile-shape sweep for the fused resize+chain kernel (BASELINE config #2).  Tuning aid.
for t in "1024 16" "1024 8" "512 16"; do
  set -- $t
  KC_RESIZE_TILE_W=$1 KC_RESIZE_TILE_H=$2 python bench.py --workload resize_blend --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 x $2', d['roofline']['kernel_us'], d['roofline']['kernel'])"
done
