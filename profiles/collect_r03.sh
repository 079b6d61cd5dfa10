#!/bin/bash
# Regenerates the round-3 measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/collect_r03.sh'    then copy gpurun_out/r03/* into profiles/
# Bench lines (HIP-event timing inside bench.py) for every BASELINE config, the rocprofv3 kernel-trace summaries of the
# default bench command and of config #2, per-kernel durations, and the PMC captures bench.py quotes as `traffic`.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
python bench.py > $OUT/r03_bench_default.json 2> $OUT/bench_default.err
echo "default done"
python bench.py --workload mix1 > $OUT/r03_bench_mix1.json 2>/dev/null
python bench.py --workload resize_blend > $OUT/r03_bench_resize_blend.json 2>/dev/null
python bench.py --workload fanin --steps 50 --warmup 5 > $OUT/r03_bench_fanin.json 2>/dev/null
python bench.py --workload chain32 --size 8192 --steps 50 --no-cpu-baseline --no-extras > $OUT/r03_bench_chain32_8192.json 2>/dev/null
python bench.py --gpus 1 --workload chain32_rows --size 8192 --steps 50 > $OUT/r03_bench_chain32_rows_8192.json 2>/dev/null
python bench.py --size 256 --steps 2000 --warmup 50 --no-cpu-baseline --no-extras > $OUT/r03_bench_chain32_256.json 2>/dev/null
python bench.py --size 1024 --steps 1000 --warmup 50 --no-cpu-baseline --no-extras > $OUT/r03_bench_chain32_1024.json 2>/dev/null
KC_CACHE_POLICY=0 python bench.py --no-cpu-baseline > $OUT/r03_bench_default_policy_off.json 2>/dev/null
KC_CACHE_POLICY=0 python bench.py --workload resize_blend --no-cpu-baseline > $OUT/r03_bench_resize_blend_policy_off.json 2>/dev/null
echo "benches done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 $R/bench.py --no-cpu-baseline > $OUT/prof_default.log 2>&1
f=$(find $OUT/prof_default -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/r03_bench_default_kernel_stats.csv
rm -rf $OUT/prof_default
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rb -- python3 $R/bench.py --workload resize_blend --no-cpu-baseline > $OUT/prof_rb.log 2>&1
f=$(find $OUT/prof_rb -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/r03_bench_resize_blend_kernel_stats.csv
rm -rf $OUT/prof_rb
echo "profiles done"
cd $R
# PMC captures (separate passes per counter group, kernel-trace only)
export KC_CAPTURE_NOTE="round 3, one MI355X via gpurun, bench.py --steps 20 --warmup 3 --no-extras"
rm -rf $R/gpurun_out/pmc; bash profiles/run_pmc.sh > $OUT/pmc_default.log 2>&1
python3 profiles/pmc_to_json.py $R/gpurun_out/pmc $OUT/r03_pmc_chain_kernel.json > /dev/null 2>> $OUT/pmc_default.log
rm -rf $R/gpurun_out/pmc; BENCH_EXTRA="--workload resize_blend" bash profiles/run_pmc.sh > $OUT/pmc_rb.log 2>&1
python3 profiles/pmc_to_json.py $R/gpurun_out/pmc $OUT/r03_pmc_upsample_chain_kernel.json 405798912 "bench.py --workload resize_blend (512^2 -> 4096^2 Triangle + 3-node blend chain), warm-up dispatches dropped" kc_upchain_ > /dev/null 2>> $OUT/pmc_rb.log
rm -rf $R/gpurun_out/pmc
(KC_SPECIALIZE=2 bash profiles/kernel_times.sh spec; KC_SPECIALIZE=0 KC_CHAIN1=0 bash profiles/kernel_times.sh interp; KC_CACHE_POLICY=0 bash profiles/kernel_times.sh policy_off) > $OUT/r03_kernel_times.txt 2>&1
bash profiles/resize_kernel_times.sh > $OUT/r03_resize_kernel_times.txt 2>&1
MODES="0 1" bash profiles/down_ab.sh 2>/dev/null > $OUT/r03_down2_ab.txt
for s in 256 1024; do python bench.py --workload fanin --size $s --steps 300 --warmup 20 --no-cpu-baseline > $OUT/r03_bench_fanin_$s.json 2>/dev/null; done
ls -la $OUT
