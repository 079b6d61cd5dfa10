#!/bin/bash
# Runs the CPU-side tests of the product library with its HOST code (runtime, graph, JSON, PNG, C API)
# (runtime, graph, JSON, PNG, specialiser, partitioner, band planner, C API)
# built under AddressSanitizer + UndefinedBehaviorSanitizer.  Device code is the regular build: GPU
# sanitizers are not available on the target pool.  Restores the regular library afterwards.
# On a GPU box the same host-only instrumentation can run the evaluator for real:
#   KC_SANITIZE_TESTS=tests/test_gpu_fuzz_graphs.py tools/sanitize_host.sh -m gpu
set -e
cd "$(dirname "$0")/.."
CL=/opt/rocm/lib/llvm/bin/clang++
SAN=${KC_SANITIZER:-address,undefined}   # or: thread
OUT=/tmp/kc_asan
mkdir -p $OUT
python -m kanter_core_amd.build >/dev/null
for f in runtime ops resize graph json png specialize partition bands comm replay u8pipe c_api; do
  $CL -x c++ -O1 -g -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -fsanitize=$SAN -fno-omit-frame-pointer \
      -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Ikanter_core_amd/csrc -c kanter_core_amd/csrc/$f.cpp -o $OUT/$f.o &
done
wait
$CL -shared -fPIC -fsanitize=$SAN -shared-libsan -o $OUT/libkanter_core_amd.so kanter_core_amd/csrc/build/kernels.o \
    kanter_core_amd/csrc/build/chain1.o kanter_core_amd/csrc/build/down2.o \
    kanter_core_amd/csrc/build/jit_texts.o $OUT/{runtime,ops,resize,graph,json,png,specialize,partition,bands,comm,replay,u8pipe,c_api}.o -L/opt/rocm/lib -lamdhip64 -lz -ldl -Wl,-rpath,/opt/rocm/lib
cp kanter_core_amd/libkanter_core_amd.so $OUT/regular.so
cp $OUT/libkanter_core_amd.so kanter_core_amd/libkanter_core_amd.so
trap 'cp $OUT/regular.so kanter_core_amd/libkanter_core_amd.so' EXIT
if [ "$SAN" = thread ]; then RT=$($CL -print-file-name=libclang_rt.tsan-x86_64.so); else RT=$($CL -print-file-name=libclang_rt.asan-x86_64.so); fi
if [ -n "${KC_SANITIZE_CMD:-}" ]; then
  # any command instead of pytest, e.g. KC_SANITIZER=thread KC_SANITIZE_CMD="python profiles/soak_threads.py 4 100"
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
      TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0${KC_TSAN_LOG:+:log_path=$KC_TSAN_LOG} $KC_SANITIZE_CMD
  exit $?
fi
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0${KC_TSAN_LOG:+:log_path=$KC_TSAN_LOG} \
    python -m pytest ${KC_SANITIZE_TESTS:-tests/test_host_graph.py tests/test_host_fuzz.py tests/test_cabi_null_args.py tests/test_cabi_symbols.py tests/test_multi_gpu_gloo.py tests/test_bands_host.py tests/test_specialize_host.py tests/test_upsample_host.py tests/test_down2_host.py tests/test_kernel_cache_host.py} -x -q "$@"
