#!/bin/bash
# A/B of two builds of the library on the same box: profiles/ab_chain.sh old.so new.so  (alternating runs)
P='import sys,json; d=json.loads(sys.stdin.read()); print("%s nodes=%s kernel_us=%.2f" % (sys.argv[1], sys.argv[2], d["roofline"]["kernel_us"]))'
for rep in 1 2 3; do
  for n in 2 16 32 64; do
    for v in "$1" "$2"; do
      cp $v kanter_core_amd/libkanter_core_amd.so
      python bench.py --nodes $n --no-cpu-baseline --no-extras --steps 200 2>/dev/null | python -c "$P" $(basename $v) $n
    done
  done
done
