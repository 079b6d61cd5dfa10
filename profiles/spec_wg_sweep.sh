#!/bin/bash
# Workgroup size of the run-time specialised chain kernel on the headline workload (and the interpreter for
# reference), alternating runs on one box.  Usage (on the GPU box): bash profiles/spec_wg_sweep.sh > gpurun_out/spec_wg_sweep.txt
for rep in 1 2; do
  for wg in 64 128 256 1024; do
    echo -n "rep $rep wg $wg: "
    KC_SPEC_WG=$wg python bench.py --no-cpu-baseline --no-extras --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel_us'], r['frac'], r['step_us_median'], r['step_us_min'], r['specialized_kernel'])"
  done
  echo -n "rep $rep interpreter: "
  KC_SPECIALIZE=0 python bench.py --no-cpu-baseline --no-extras --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel_us'], r['frac'], r['step_us_median'], r['step_us_min'], r['specialized_kernel'])"
done
