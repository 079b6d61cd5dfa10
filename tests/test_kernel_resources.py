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
    table = {}
    tmp = tmp_path_factory.mktemp("res")
    for unit in ("kernels.hip", "chain1.hip", "down2.hip"):
        src = os.path.join(ROOT, "kanter_core_amd", "csrc", unit)
        cmd = [hipcc] + kbuild.FLAGS + kbuild.DEVICE_FLAGS + ["-x", "hip", "-Rpass-analysis=kernel-resource-usage", "-c", src,
                                                              "-o", str(tmp / (unit + ".o"))]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
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
        if "chain_kernel" in name and "ILi" in name and ("Li2EEEv" in name or "Li2ELb" in name):
            continue  # MODE 2 calls the f64 pow routine: a call frame is expected
        assert u.get("ScratchSize", 0) == 0, (name, u)


def test_lean_chain_kernels_keep_full_occupancy(usage):
    # chain_kernel<K, 4, 0, NT>: K = 1, 2 must allow 8 waves per SIMD (<= 64 VGPRs), with and without cache hints
    for nt in (0, 1):
        for k in (1, 2):
            (u,) = find(usage, "chain_kernelILi%dELi4ELi0ELb%dEEE" % (k, nt))
            assert u["VGPRs"] <= 64, (k, nt, u)
        (u,) = find(usage, "chain_kernelILi4ELi4ELi0ELb%dEEE" % nt)
        assert u["VGPRs"] <= 128, (nt, u)


def test_resize_kernels_fit_their_budgets(usage):
    for frag in ("resize_lds_kernelILi2ELi3EEE", "resize_chain_kernelILi2ELi3EEE", "resize_down_kernelILi4EEE",
                 "resize_poly_kernelILi6ELi4EEE"):
        (u,) = find(usage, frag)
        assert u["VGPRs"] <= 128, (frag, u)
    (u,) = find(usage, "resize_poly_kernelILi2ELi8EEE")
    assert u["VGPRs"] <= 160, u  # three waves per SIMD (Triangle 8:1 has 688 waves for 1024 SIMDs)
    # resize_poly_kernel keeps its A * RT weights as vector register pairs since round 4 (48 registers at A = 6, RT = 8, next to
    # two trips of eight 16-byte rows): two waves per SIMD are what its launches can use (688 - 1376 waves on 1024 SIMDs);
    # no variant may read weights back from lanes again (v_readlane: what the scalar-register form cost)
    (u,) = find(usage, "resize_poly_kernelILi6ELi8EEE")
    assert u["VGPRs"] <= 256, u


def test_poly2_kernels_fit_their_budgets(usage):
    # resize_poly2_kernel<A, RT>: 8-byte lanes, A * RT / 2 weight pairs + one or two trips of RT rows: four waves per SIMD at the
    # widest instantiation, and its barrier must stay the bare one (no vmcnt(0) in front of it would show up as time, not here)
    for frag, budget in (("resize_poly2_kernelILi6ELi8EEE", 128), ("resize_poly2_kernelILi4ELi8EEE", 128), ("resize_poly2_kernelILi6ELi4EEE", 128),
                         ("resize_poly2_kernelILi6ELi2EEE", 96), ("resize_poly2_kernelILi2ELi8EEE", 96)):
        (u,) = find(usage, frag)
        assert u["VGPRs"] <= budget, (frag, u)


def test_down2_kernels_fit_their_budgets(usage):
    # resize_down2_kernel<HC, NW4, ONE> (down2.hip): 16 source rows in flight per lane (64 VGPRs) next to 8 packed sums.  The
    # single-chunk forms must allow 4 waves per SIMD with three columns per lane (Lanczos3 at ratios below 1.6) and 5 otherwise;
    # nothing may spill, and a workgroup's four wave-private transposition areas are 18 KB of LDS (8 workgroups per CU).
    seen = 0
    for name, u in usage.items():
        if "resize_down2_kernel" not in name:
            continue
        seen += 1
        assert u.get("ScratchSize", 0) == 0, (name, u)
        assert u["VGPRs"] <= 128, (name, u)
        assert u["LDS"] <= 20 * 1024, (name, u)
        if "Lb1EEE" in name:  # single-chunk forms
            assert u["VGPRs"] <= (112 if "kernelILi3E" in name else 96), (name, u)
    assert seen == 16


def test_upsample_kernels_fit_their_budgets(usage):
    # the plain integer-ratio up-sampling kernel (Triangle, wide tiles): 8 waves per SIMD
    # (template arguments: taps, wide tile, nontemporal stores, ratio 2 = half quads)
    for frag in ("upsample_kernelILi3ELb1ELb0ELb0EEE", "upsample_kernelILi3ELb1ELb1ELb0EEE", "upsample_kernelILi1ELb1ELb0ELb0EEE",
                 "upsample_kernelILi3ELb1ELb0ELb1EEE", "upsample_kernelILi3ELb1ELb1ELb1EEE"):
        (u,) = find(usage, frag)
        assert u["VGPRs"] <= 64, (frag, u)
    for frag in ("upsample_kernelILi7ELb1ELb0ELb0EEE", "upsample_kernelILi5ELb0ELb0ELb0EEE", "upsample_kernelILi7ELb1ELb0ELb1EEE",
                 "upsample_kernelILi5ELb1ELb1ELb1EEE"):
        (u,) = find(usage, frag)
        assert u["VGPRs"] <= 96, (frag, u)
    # the interpreter-driven fused form (first sightings only): what resize_chain_kernel<2,3> needed 108 for
    for frag in ("upsample_chain_kernelILi2ELi3ELb1ELb0EEE", "upsample_chain_kernelILi2ELi3ELb1ELb1EEE"):
        (u,) = find(usage, frag)
        assert u["VGPRs"] <= 96, (frag, u)
    for name, u in usage.items():
        if "chain1_kernel" in name or "upsample" in name:
            assert u.get("ScratchSize", 0) == 0, (name, u)


def test_specialised_upsample_chain_kernel_keeps_full_occupancy(tmp_path):
    """Config #2's program inside the up-sampling kernel, as the run-time specialiser emits it (generated source compiled
    here with hipcc and the parity flags): <= 64 VGPRs = 8 waves per SIMD, no scratch.  (Round 2's resize_chain_kernel<2,3>:
    108 VGPRs, 4 waves per SIMD.)"""
    import kanter_core_amd as kc
    from kanter_core_amd import build as kbuild
    hipcc = kbuild._hipcc()
    if shutil.which(hipcc) is None and not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    word = lambda code, src: code | ((src + 1) << 8)  # noqa: E731
    src = kc.specialize_compile_check_upsample([word(0, 1), word(3, 0), word(1, 1)], n_in=2, start_src=0, taps=3, wide=True)
    f = tmp_path / "upchain.hip"
    f.write_text("#include <hip/hip_runtime.h>\n" + src)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", str(f), "-o", str(tmp_path / "upchain.o")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    vg = [int(x) for x in re.findall(r"remark:\s+VGPRs: (\d+)", r.stdout)]
    sc = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stdout)]
    assert vg and max(vg) <= 64, r.stdout[-2000:]
    assert sc and max(sc) == 0
