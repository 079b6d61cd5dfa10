"""Build-time guard for the kernels' register / scratch budgets (hipcc cross-compiles without a
GPU).  The chain kernel's speed depends on two code-generation facts that a source or compiler
change can silently break: its step table must be read through scalar loads (no scratch copy of
the kernel-argument block), and the {+,-,*} instantiation must fit 64 VGPRs (8 waves per SIMD)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    from kanter_core_amd import build as kbuild
    hipcc = kbuild._hipcc()
    if shutil.which(hipcc) is None and not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("res") / "kernels.o"
    src = os.path.join(ROOT, "kanter_core_amd", "csrc", "kernels.hip")
    cmd = [hipcc] + kbuild.FLAGS + kbuild.DEVICE_FLAGS + ["-x", "hip", "-Rpass-analysis=kernel-resource-usage", "-c", src,
                                                          "-o", str(out)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    table = {}
    name = None
    for line in r.stdout.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            table[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            table[name][m.group(1).split()[0]] = int(m.group(2))
    return table


def find(usage, fragment):
    hits = [v for k, v in usage.items() if fragment in k]
    assert hits, fragment
    return hits


def test_no_kernel_spills_to_scratch(usage):
    for name, u in usage.items():
        if "kc" in name and "ILi" in name and "Li2EEEv" in name and "chain_kernel" in name:
            continue  # MODE 2 calls the f64 pow routine: a call frame is expected
        assert u.get("ScratchSize", 0) == 0, (name, u)


def test_lean_chain_kernels_keep_full_occupancy(usage):
    # chain_kernel<K, 4, 0>: K = 1, 2 must allow 8 waves per SIMD (<= 64 VGPRs)
    for k in (1, 2):
        (u,) = find(usage, "chain_kernelILi%dELi4ELi0EEE" % k)
        assert u["VGPRs"] <= 64, (k, u)
    (u,) = find(usage, "chain_kernelILi4ELi4ELi0EEE")
    assert u["VGPRs"] <= 128, u


def test_resize_kernels_fit_their_budgets(usage):
    for frag in ("resize_lds_kernelILi2ELi3EEE", "resize_chain_kernelILi2ELi3EEE", "resize_down_kernelILi4EEE",
                 "resize_poly_kernelILi6ELi4EEE", "resize_poly_kernelILi6ELi8EEE", "resize_poly_kernelILi2ELi8EEE"):
        (u,) = find(usage, frag)
        assert u["VGPRs"] <= 128, (frag, u)
