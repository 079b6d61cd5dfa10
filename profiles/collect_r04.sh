#!/bin/bash
# Regenerates the round-4 measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/collect_r04.sh'    then copy gpurun_out/r04/* into profiles/
# Bench lines of every BASELINE config (HIP-event timing inside bench.py), TWO rocprofv3 kernel-trace summaries per headline
# workload -- the contract's timed region alone (--no-extras: the warm figure, frac) and the cold rotation alone (--cold:
# frac_cold) -- each reproducing its fraction from bytes / AverageNs, the PMC capture bench.py quotes as `traffic`, and the
# per-kernel durations of the streaming and resampling kernels.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04
mkdir -p $OUT
cd $R
python bench.py > $OUT/r04_bench_default.json 2> $OUT/bench_default.err
echo "default done"
python bench.py --workload mix1 > $OUT/r04_bench_mix1.json 2>/dev/null
python bench.py --workload resize_blend > $OUT/r04_bench_resize_blend.json 2>/dev/null
python bench.py --workload fanin --steps 50 --warmup 5 > $OUT/r04_bench_fanin.json 2>/dev/null
python bench.py --workload e2e --steps 100 --warmup 6 > $OUT/r04_bench_e2e.json 2>/dev/null
python bench.py --workload chain32 --size 8192 --steps 50 --no-cpu-baseline --no-extras > $OUT/r04_bench_chain32_8192.json 2>/dev/null
python bench.py --gpus 1 --workload chain32_rows --size 8192 --steps 50 > $OUT/r04_bench_chain32_rows_8192.json 2>/dev/null
python bench.py --size 256 --steps 2000 --warmup 50 --no-cpu-baseline --no-extras > $OUT/r04_bench_chain32_256.json 2>/dev/null
python bench.py --size 1024 --steps 1000 --warmup 50 --no-cpu-baseline --no-extras > $OUT/r04_bench_chain32_1024.json 2>/dev/null
for s in 256 1024; do python bench.py --workload fanin --size $s --steps 300 --warmup 20 --no-cpu-baseline > $OUT/r04_bench_fanin_$s.json 2>/dev/null; done
# a process that has to compile (empty cache, no packaged kernels): what the first sightings cost, what the driver would have seen
d=$(mktemp -d); mv $R/kanter_core_amd/kernel_cache $d/packaged
KC_KERNEL_CACHE_DIR=$d/user python bench.py --no-cpu-baseline --no-extras > $OUT/r04_bench_default_empty_cache.json 2>/dev/null
KC_KERNEL_CACHE_DIR=$d/user python bench.py --no-cpu-baseline --no-extras > $OUT/r04_bench_default_second_process.json 2>/dev/null
mv $d/packaged $R/kanter_core_amd/kernel_cache; rm -rf $d
echo "benches done"
cd /tmp && export TMPDIR=/tmp
trace() {  # name, bench args...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/prof_$name.log 2>&1
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/r04_${name}_kernel_stats.csv
  grep '^{' $OUT/prof_$name.log > $OUT/r04_${name}_bench_line.json
  rm -rf $OUT/prof_$name $OUT/prof_$name.log
}
trace default_warm --no-extras
trace default_cold --cold
trace resize_blend_warm --workload resize_blend --no-extras
trace resize_blend_cold --workload resize_blend --cold
trace mix1_warm --workload mix1 --no-extras
trace mix1_cold --workload mix1 --cold
trace fanin --workload fanin --steps 50 --warmup 5
trace e2e --workload e2e --steps 100 --warmup 6
echo "profiles done"
cd $R
# PMC captures (separate passes per counter group, kernel-trace only)
export KC_CAPTURE_NOTE="round 4, one MI355X via gpurun, bench.py --steps 20 --warmup 3 --no-extras"
rm -rf $R/gpurun_out/pmc; bash profiles/run_pmc.sh > $OUT/pmc_default.log 2>&1
python3 profiles/pmc_to_json.py $R/gpurun_out/pmc $OUT/r04_pmc_chain_kernel.json > /dev/null 2>> $OUT/pmc_default.log
rm -rf $R/gpurun_out/pmc; BENCH_EXTRA="--workload resize_blend" bash profiles/run_pmc.sh > $OUT/pmc_rb.log 2>&1
python3 profiles/pmc_to_json.py $R/gpurun_out/pmc $OUT/r04_pmc_upsample_chain_kernel.json 405798912 "bench.py --workload resize_blend (512^2 -> 4096^2 Triangle + 3-node blend chain), warm-up dispatches dropped" kc_upchain_ > /dev/null 2>> $OUT/pmc_rb.log
rm -rf $R/gpurun_out/pmc
(KC_SPECIALIZE=2 bash profiles/kernel_times.sh spec) > $OUT/r04_kernel_times.txt 2>&1
bash profiles/resize_kernel_times.sh > $OUT/r04_resize_kernel_times.txt 2>&1
ls -la $OUT
