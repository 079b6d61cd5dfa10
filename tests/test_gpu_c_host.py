"""GPU parity through the C ABI from a plain C program: tests/c_host/host_check.c is compiled with gcc against
include/kanter_core_amd.h, linked to libkanter_core_amd.so (and, being a test, to the oracle) and run as its
own process -- no Python, ctypes or torch between the caller and the library, which is how a Rust
`extern "C"` binding would use it."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_program_matches_the_oracle(tmp_path):
    import kanter_core_amd  # noqa: F401  (makes sure the library is built)
    from oracle import oracle as orc
    orc.lib()  # builds oracle/libkc_oracle.so if needed
    exe = str(tmp_path / "host_check")
    libdir, ordir = os.path.join(ROOT, "kanter_core_amd"), os.path.join(ROOT, "oracle")
    subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", os.path.join(ROOT, "tests", "c_host", "host_check.c"),
                    "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lkanter_core_amd", "-L", ordir, "-lkc_oracle",
                    "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath," + ordir, "-o", exe], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert "c host: 0 failures" in r.stdout, r.stdout
