#!/bin/bash
# Regenerates the round-1 measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'profiles/collect_r01.sh'    then copy gpurun_out/r01/* into profiles/
# Bench lines (HIP-event timing inside bench.py), rocprofv3 kernel-trace summaries of the same commands,
# the per-kernel microbenchmark and the true kernel durations of the resize cases.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r01
mkdir -p $OUT
cd $R
python bench.py > $OUT/r01_bench_default.json 2> $OUT/bench_default.err
python bench.py --workload mix1 --no-cpu-baseline --no-extras > $OUT/r01_bench_mix1.json 2>/dev/null
python bench.py --workload resize_blend --no-cpu-baseline --no-extras > $OUT/r01_bench_resize_blend.json 2>/dev/null
python bench.py --workload chain32 --size 8192 --steps 50 --no-cpu-baseline --no-extras > $OUT/r01_bench_chain32_8192.json 2>/dev/null
python bench.py --workload fanin --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/r01_bench_fanin.json 2>/dev/null
python profiles/kernel_microbench.py --reps 50 > $OUT/r01_kernel_microbench.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for w in chain32 resize_blend; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python3 $R/bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/prof_$w.log 2>&1
  f=$(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/r01_bench_${w}_kernel_stats.csv
  rm -rf $OUT/prof_$w
done
cd $R && profiles/resize_kernel_times.sh > $OUT/r01_resize_kernel_times.txt 2>&1
ls -la $OUT
