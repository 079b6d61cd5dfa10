#!/bin/bash
# HeightToNormal with the plain workgroup -> pixel mapping (KC_H2N_TILED=0) and the XCD-consistent tiles (default) on square
# planes of several sizes: kernel-trace durations per size.     gpurun -- 'bash profiles/h2n_ab.sh'
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/h2n_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in 0 default 1; do
  rm -rf $OUT/t_$m
  if [ $m = default ]; then env=""; else env="KC_H2N_TILED=$m"; fi
  env $env true
  ( [ $m = default ] || export KC_H2N_TILED=$m; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$m -- python3 $R/profiles/h2n_sizes.py 20 > $OUT/t_$m.log 2>&1 )
  f=$(find $OUT/t_$m -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$m" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "height_to_normal" in r["Kernel_Name"]:
        key = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]), r["Kernel_Name"].split("<")[1].split(">")[0])
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (g, t), v in sorted(d.items()):
    v = v[2:] if len(v) > 4 else v
    px = g * 4  # one quad per thread
    print("KC_H2N_TILED=%-7s threads=%9d <%s>  avg=%7.1f us min=%7.1f us  ~%.2f of 8 TB/s" % (sys.argv[2], g, t, sum(v) / len(v), min(v), 16.0 * px / (sum(v) / len(v)) / 8e6))
PY
done
