#!/bin/bash
# Config #2 (bench.py --workload resize_blend) under a kernel trace, once per setting of the environment given as arguments:
#   gpurun -- 'bash profiles/config2_ab.sh "KC_RESIZE_MODE=4" "KC_RESIZE_TILE_H=16" ""'
# Prints, per setting, the bench line's kernel time (HIP events) and the kernel-trace averages of the resize / chain kernels.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/config2_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for setting in "$@"; do
  i=$((i+1))
  echo "== setting $i: '$setting'"
  ( [ -n "$setting" ] && export $setting
    python3 $R/bench.py --workload resize_blend --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('events: kernel_us=%.2f frac=%.3f median=%.2f min=%.2f parity=%s spec=%s' % (r['kernel_us'], r['frac'], r['step_us_median'], r['step_us_min'], d.get('parity'), r.get('specialized_kernel')))"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$i -- python3 $R/bench.py --workload resize_blend --steps 100 --warmup 20 --no-cpu-baseline --no-extras > $OUT/p$i.log 2>&1
    f=$(find $OUT/p$i -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('upsample','resize','kc_chain','kc_up','chain_kernel')):
        print('trace: %-70s calls=%s avg_us=%.2f min_us=%.2f' % (n[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
    rm -rf $OUT/p$i )
done
