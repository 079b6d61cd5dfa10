#!/bin/bash
# A/B of environment settings on one bench workload, alternating: gpurun -- 'bash profiles/ab_env.sh "<bench args>" "KC_CACHE_POLICY=0" "KC_CACHE_POLICY=1"'
set -u
R=$GRAFT_REPO_ROOT
ARGS=$1; shift
P='import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]
print("%-24s value=%.0f kernel_us=%.2f frac=%.3f median=%.2f min=%.2f parity=%s" % (sys.argv[1], d["value"], r["kernel_us"], r["frac"], r["step_us_median"], r["step_us_min"], d.get("parity")))'
for rep in 1 2; do
  for setting in "$@"; do
    ( [ -n "$setting" ] && export $setting; python3 $R/bench.py $ARGS 2>/dev/null | python3 -c "$P" "$setting" )
  done
done
