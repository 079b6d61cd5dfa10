#!/bin/bash
# Regenerates the round-2 measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/collect_r02.sh'    then copy gpurun_out/r02/* into profiles/
# Bench lines (HIP-event timing inside bench.py) for every BASELINE config, the rocprofv3 kernel-trace summary of the
# default bench command, per-kernel durations and the resize cases (durations + HBM traffic by PMC).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd $R
python bench.py > $OUT/r02_bench_default.json 2> $OUT/bench_default.err
echo "default done"
python bench.py --workload mix1 > $OUT/r02_bench_mix1.json 2>/dev/null
python bench.py --workload resize_blend > $OUT/r02_bench_resize_blend.json 2>/dev/null
python bench.py --workload fanin --steps 50 --warmup 5 > $OUT/r02_bench_fanin.json 2>/dev/null
python bench.py --workload chain32 --size 8192 --steps 50 --no-cpu-baseline --no-extras > $OUT/r02_bench_chain32_8192.json 2>/dev/null
python bench.py --workload chain32_rows --size 8192 --steps 50 --no-cpu-baseline > $OUT/r02_bench_chain32_rows_8192.json 2>/dev/null
python bench.py --size 256 --steps 2000 --warmup 50 --no-cpu-baseline --no-extras > $OUT/r02_bench_chain32_256.json 2>/dev/null
echo "benches done"
python profiles/kernel_microbench.py --reps 50 > $OUT/r02_kernel_microbench.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 $R/bench.py --no-cpu-baseline > $OUT/prof_default.log 2>&1
f=$(find $OUT/prof_default -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/r02_bench_default_kernel_stats.csv
rm -rf $OUT/prof_default
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rb -- python3 $R/bench.py --workload resize_blend --no-cpu-baseline > $OUT/prof_rb.log 2>&1
f=$(find $OUT/prof_rb -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/r02_bench_resize_blend_kernel_stats.csv
rm -rf $OUT/prof_rb
echo "profiles done"
cd $R
(KC_SPECIALIZE=2 bash profiles/kernel_times.sh spec; python3 profiles/split_chain_times.py gpurun_out/kt_spec; KC_SPECIALIZE=0 bash profiles/kernel_times.sh interp > /dev/null; python3 profiles/split_chain_times.py gpurun_out/kt_interp) > $OUT/r02_kernel_times.txt 2>&1
timeout -k 10 500 bash profiles/resize_kernel_times.sh > $OUT/r02_resize_kernel_times.txt 2>&1
timeout -k 10 500 bash profiles/resize_traffic.sh > $OUT/r02_resize_traffic.txt 2>&1
ls -la $OUT
