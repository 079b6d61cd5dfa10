#!/bin/bash
# Runs the oracle-vs-golden tests with the C oracle built under AddressSanitizer + UBSan (CPU only;
# GPU sanitizers are not available on the target pool).  Restores the regular build afterwards.
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -fPIC -std=c11 -fno-fast-math -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o /tmp/libkc_oracle_asan.so oracle/kc_oracle.c -lm
make -C oracle -B libkc_oracle.so >/dev/null
cp oracle/libkc_oracle.so /tmp/libkc_oracle.regular
cp /tmp/libkc_oracle_asan.so oracle/libkc_oracle.so
trap 'cp /tmp/libkc_oracle.regular oracle/libkc_oracle.so' EXIT
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle_golden.py -x -q
