#!/bin/bash
# Collects per-kernel PMC counters for the headline bench (run on the GPU box via gpurun).
# Counters go in separate passes (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_EXTRA:-}"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $ARGS > $OUT/pass$i.log 2>&1
  echo "pass $i ($set) exit $?"
done
find $OUT -name "*counter_collection.csv" | head
