P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "kernel_us=%.2f frac=%.3f" % (r["kernel_us"], r["frac"]))'
cp kanter_core_amd/libkanter_core_amd.so /tmp/ru4.so
for rep in 1 2; do for v in /tmp/ru4.so profiles/ab_libs/ru2.so profiles/ab_libs/ru1.so; do cp $v kanter_core_amd/libkanter_core_amd.so; python bench.py --workload resize_blend --no-cpu-baseline --steps 200 2>/dev/null | python -c "$P" $(basename $v); done; done
cp /tmp/ru4.so kanter_core_amd/libkanter_core_amd.so
