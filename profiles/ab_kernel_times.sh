#!/bin/bash
# kernel_times.sh under alternating library variants: gpurun -- 'bash profiles/ab_kernel_times.sh <grep pattern> base variant ...'
set -u
R=$GRAFT_REPO_ROOT
PAT=$1; shift
cp $R/kanter_core_amd/libkanter_core_amd.so /tmp/base.so
trap 'cp /tmp/base.so $R/kanter_core_amd/libkanter_core_amd.so' EXIT  # also on a timeout or Ctrl-C
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then cp /tmp/base.so $R/kanter_core_amd/libkanter_core_amd.so; else cp $R/profiles/ab_libs/$v.so $R/kanter_core_amd/libkanter_core_amd.so; fi
    echo "-- $v"; bash $R/profiles/kernel_times.sh ab_$v 30 2>&1 | grep -E "$PAT"
  done
done
cp /tmp/base.so $R/kanter_core_amd/libkanter_core_amd.so
