"""The kernel cache without a device: hiprtc cross-compiles, so the build can pre-compile the BASELINE programs
(kanter_core_amd/baseline_programs.jsonl -> kanter_core_amd/kernel_cache/) and these tests can check the file format."""
import json
import os
import struct

import pytest

import kanter_core_amd as kc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MANIFEST = os.path.join(ROOT, "kanter_core_amd", "baseline_programs.jsonl")
PACKAGED = os.path.join(ROOT, "kanter_core_amd", "kernel_cache")


def test_precompile_writes_a_wellformed_file(tmp_path):
    words = [0 | (2 << 8), 3 | (2 << 8), 2 | (1 << 8)]  # in0 + in1, * in1, in0 - acc
    kc.kernel_cache_precompile(words, 2, 0, True, 0, tmp_path)
    files = [f for f in os.listdir(tmp_path) if f.endswith(".kcco")]
    assert len(files) == 1 and files[0].startswith("kc_chain_") and os.path.exists(os.path.join(tmp_path, ".populated"))
    blob = open(os.path.join(tmp_path, files[0]), "rb").read()
    assert blob[:8] == b"KCCO0001"
    key_len, _, code_len, _ = struct.unpack("<4Q", blob[8:40])
    assert len(blob) == 40 + key_len + code_len and blob[40 + key_len:40 + key_len + 4] == b"\x7fELF"
    # the key holds the signature by value: n_in, n_ops, start_src, 'f'lat, ..., the words
    key = blob[40:40 + key_len]
    assert struct.unpack("<IIi", key[:12]) == (2, 3, 0) and key[12:13] == b"f" and key.endswith(struct.pack("<3I", *words))
    # a different cache-policy mask is a different kernel (another file), the same program again is the same file
    kc.kernel_cache_precompile(words, 2, 0, True, 0x101, tmp_path)
    kc.kernel_cache_precompile(words, 2, 0, True, 0, tmp_path)
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".kcco")]) == 2
    assert open(os.path.join(tmp_path, files[0]), "rb").read() == blob  # deterministic compile
    with pytest.raises(kc.TexProError):
        kc.kernel_cache_precompile([99], 1, 0, True, 0, tmp_path)  # unknown step code


@pytest.mark.skipif(not os.path.exists(MANIFEST), reason="no manifest of BASELINE programs recorded yet")
def test_the_build_has_precompiled_every_baseline_program():
    entries = [json.loads(line) for line in open(MANIFEST) if line.strip()]
    assert entries
    from kanter_core_amd import build as kbuild
    kbuild.populate_kernel_cache()
    have = os.listdir(PACKAGED)
    for e in entries:
        assert any(f.startswith(e["name"] + "-") and f.endswith(".kcco") for f in have), e["name"]
