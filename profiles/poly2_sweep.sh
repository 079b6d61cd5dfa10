#!/bin/bash
# resize_poly2_kernel: trips in flight (variant libraries of tools/build_variant.sh) x band height x XCD order, and the in-kernel clocks.
#   gpurun -- 'bash profiles/poly2_sweep.sh'
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
run() { # lib rows xcd
  unset KC_LIB_PATH KC_POLY_ROWS KC_POLY_XCD
  [ -n "$1" ] && export KC_LIB_PATH=$R/profiles/ab_libs/$1.so
  [ -n "$2" ] && export KC_POLY_ROWS=$2
  [ -n "$3" ] && export KC_POLY_XCD=$3
  echo "== lib=${1:-default(nb2)} rows=${2:-auto} xcd=${3:-default}"
  bash $R/profiles/poly_cases.sh 1
}
run "" "" ""
run "" "" 0
run "" 24 ""
run "" 8 ""
run p2_nb1 "" ""
run p2_nb3 "" ""
unset KC_LIB_PATH KC_POLY_ROWS KC_POLY_XCD
echo "== 4 planes"
bash $R/profiles/poly_cases.sh 4
echo "== clocks"
for c in "4096 512 Gaussian" "4096 1024 Lanczos3" "4096 2048 Lanczos3"; do
  KC_LIB_PATH=$R/profiles/ab_libs/p2_timing.so KC_POLY_TIMING=1 python3 $R/profiles/resize_one.py $c 3 1 2>&1 | grep "timing" | tail -1
done
