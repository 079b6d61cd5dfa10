#!/bin/bash
# A/B of library variants built by tools/build_variant.sh on one workload, alternating, HIP-event kernel time + parity:
#   gpurun -- 'bash profiles/ab_libs.sh resize_blend base hard_ru4 ru2'     ("base" = the shipped library)
set -u
R=$GRAFT_REPO_ROOT
W=$1; shift
cp $R/kanter_core_amd/libkanter_core_amd.so /tmp/base.so
# whatever ends this script -- the last line, a timeout, Ctrl-C -- the shipped library goes back in place
trap 'cp /tmp/base.so $R/kanter_core_amd/libkanter_core_amd.so' EXIT
P='import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]
print("%-12s kernel_us=%.2f frac=%.3f median=%.2f min=%.2f parity=%s" % (sys.argv[1], r["kernel_us"], r["frac"], r["step_us_median"], r["step_us_min"], d.get("parity")))'
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then cp /tmp/base.so $R/kanter_core_amd/libkanter_core_amd.so; else cp $R/profiles/ab_libs/$v.so $R/kanter_core_amd/libkanter_core_amd.so; fi
    python3 $R/bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline ${BENCH_EXTRA:-} 2>/dev/null | python3 -c "$P" $v
  done
done
cp /tmp/base.so $R/kanter_core_amd/libkanter_core_amd.so
