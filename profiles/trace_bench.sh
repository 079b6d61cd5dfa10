#!/bin/bash
# Kernel-trace per-dispatch durations of one bench.py command: bash profiles/trace_bench.sh "<bench args>" <kernel name pattern> [env ...]
set -u
R=$GRAFT_REPO_ROOT
ARGS=$1; PAT=$2; shift 2
for e in "$@"; do export $e; done
OUT=$R/gpurun_out/trace_bench_$$
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py $ARGS > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench (under the profiler): kernel_us=%.2f frac=%.3f' % (r['kernel_us'], r['frac']))
except Exception as e: print('no bench line', e)"
python3 $R/profiles/trace_durations.py $OUT "$PAT"
rm -rf $OUT
