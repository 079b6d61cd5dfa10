#!/bin/bash
# Records the signatures of the chain programs the BASELINE workloads compile (run on a GPU box; from the repo root):
#   tools/dump_baseline_programs.sh            -> gpurun_out/baseline_programs.jsonl
# then, here:  sort -u gpurun_out/baseline_programs.jsonl > kanter_core_amd/baseline_programs.jsonl
# Every run starts with an empty kernel cache (so everything is compiled in-process) and appends what it compiled to the
# manifest (KC_KERNEL_CACHE_MANIFEST, csrc/specialize.cpp).  The build then pre-compiles exactly these (kanter_core_amd/build.py).
set -e
out=gpurun_out/baseline_programs.jsonl
mkdir -p gpurun_out
rm -f "$out"
export KC_KERNEL_CACHE_MANIFEST="$PWD/$out"
run() {
    d=$(mktemp -d)
    KC_KERNEL_CACHE_DIR="$d" python bench.py --steps 6 --warmup 4 --no-cpu-baseline "$@" > /dev/null
    rm -rf "$d"
}
run                                               # the headline (its cold rotation and the PCIe leg included)
run --workload mix1
run --workload resize_blend
run --workload fanin                              # config #4 on one GPU
run --workload chain32 --size 8192 --no-extras    # config #3
for s in 256 512 1024 2048; do run --workload chain32 --size $s --no-extras; run --workload fanin --size $s; done
for s in 1024 2048; do run --workload resize_blend --size $s --no-extras; done
# row bands of config #3 as 2 / 4 / 8 ranks cut them, and of config #4 (the band plans): one process per band shape
d=$(mktemp -d)
KC_KERNEL_CACHE_DIR="$d" python tools/band_programs.py
rm -rf "$d"
d=$(mktemp -d)
KC_SPECIALIZE=2 KC_KERNEL_CACHE_DIR="$d" python -c "import __graft_entry__ as g; g.smoke()" > /dev/null   # (compile at first sight: smoke sees every program once)
rm -rf "$d"
sort -u "$out" -o "$out"
wc -l "$out"
