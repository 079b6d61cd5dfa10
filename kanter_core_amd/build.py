"""Builds kanter_core_amd/libkanter_core_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m kanter_core_amd.build [--force]

The shared library is the C-ABI drop-in declared in include/kanter_core_amd.h.  It is kept
in-tree (git-ignored) so it travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libkanter_core_amd.so")
SOURCES = ["kernels.hip", "runtime.cpp", "ops.cpp", "resize.cpp", "graph.cpp", "json.cpp", "png.cpp", "c_api.cpp"]
HEADERS = ["kc_internal.hpp", "kc_runtime.hpp", os.path.join("..", "..", "include", "kanter_core_amd.h"), "pow_positive.inc"]

# -ffp-contract=off and IEEE divide/sqrt are parity requirements (the reference's Rust loops never
# fuse a*b+c and divide exactly); see INTEGRATION.md.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-fast-math",
         "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-result"]
# Device code only: leave wave-uniform branches (the chain kernel's scalar step dispatch) as plain
# scalar branches instead of structurizing them -- removes the per-arm "Flow" blocks and register
# copies (90 -> 58 VGPRs, ~20 -> ~13 scalar instructions per step; profiles/r01_chain_unroll.md).
DEVICE_FLAGS = ["-mllvm", "-structurizecfg-skip-uniform-regions=1"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + (DEVICE_FLAGS + ["-x", "hip"] if src.endswith(".hip") else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    with ThreadPoolExecutor(max_workers=4) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
