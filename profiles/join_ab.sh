#!/bin/bash
# BASELINE config #4 on one GPU with a Mix of two unevaluated chains kept in one program (default) and run on the spot as before
# (KC_JOIN=0): bench.py --workload fanin at three sizes.      gpurun -- 'bash profiles/join_ab.sh'
cd $GRAFT_REPO_ROOT
for j in 0 1; do
  for s in 256 1024 4096; do
    KC_JOIN=$j python bench.py --workload fanin --size $s --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('KC_JOIN=$j fanin %4d^2  %8.1f us per evaluation  launches=%s  algorithmic bytes per evaluation=%.0f  frac=%.3f  parity=%s' % ($s, d['ms_per_step'] * 1e3, r.get('launches_per_step'), r.get('algorithmic_bytes_per_step'), r['frac'], (d.get('parity') or {}).get('bit_mismatches')))"
  done
done
