"""The C-ABI library loads on a machine without a GPU and exports every entry point that
include/kanter_core_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kanter_core_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"KC_API\s+[\w\s\*]+?\b(kc_\w+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    assert len(syms) > 80
    for must in ("kc_init", "kc_mix_process", "kc_resize_buffers", "kc_height_to_normal_process",
                 "kc_image_to_u8", "kc_image_from_u8", "kc_live_graph_await_clean", "kc_node_graph_from_json"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from kanter_core_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_covers_header_and_nothing_else():
    from kanter_core_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import kanter_core_amd as kc
    with pytest.raises(kc.TexProError) as e:
        kc.init(0)
    assert e.value.kind == "NoDevice"
    with pytest.raises(kc.TexProError):
        kc.SlotImage.from_planes([__import__("numpy").zeros((4, 4), "float32")])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "kanter_core_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower() or f == "build.py", f


def test_rust_binding_source_is_in_sync_with_the_header():
    """bindings/rust/kanter_core_amd_sys.rs (generated, source only: no Rust toolchain here) declares
    exactly the header's entry points."""
    import subprocess
    import sys
    rs = os.path.join(ROOT, "bindings", "rust", "kanter_core_amd_sys.rs")
    before = open(rs).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "bindings", "gen_rust.py")], stdout=subprocess.DEVNULL)
    assert open(rs).read() == before, "re-run bindings/gen_rust.py"
    declared = sorted(re.findall(r"pub fn (kc_\w+)\(", before))
    assert declared == declared_symbols()
