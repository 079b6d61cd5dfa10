#!/bin/bash
# SQ-side PMC counters (issue / wait / LDS): profiles/run_pmc_sq.sh <tag> <bench args...>
#   or, for another script: SCRIPT=profiles/resize_one.py profiles/run_pmc_sq.sh <tag> <script args...>
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "${SCRIPT:-}" ]; then ARGS="$GRAFT_REPO_ROOT/$SCRIPT $*"; else ARGS="$GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $*"; fi
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $ARGS > $OUT/pass$i.log 2>&1
  echo "pass $i ($set) exit $?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "fill" in k or "from_u8" in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
