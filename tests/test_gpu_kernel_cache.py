"""Compiled kernels outlive the process (csrc/specialize.cpp, "code objects that outlive the process"): a code object hiprtc
produced is written to the cache directory, the first sighting of the same program in a later process (here: after
kc.specialize_reset()) loads it instead of compiling, and a file that is corrupted, truncated, stale (made by another version of
the generator / compiler) or that belongs to another program is refused and replaced by a fresh compile -- never run."""
import os
import struct

import numpy as np
import pytest

from util import SEED_A, SEED_B, assert_planes, synthetic_rgba

pytestmark = pytest.mark.gpu


@pytest.fixture()
def kc(tmp_path):
    import kanter_core_amd as kc
    kc.init(0)
    kc.specialize_wait()
    kc.kernel_cache_set_dir(str(tmp_path))
    kc.specialize_reset()
    yield kc
    kc.specialize_wait()
    kc.kernel_cache_set_dir(None)
    kc.set_specialize(1, 2)
    kc.specialize_reset()


def program(kc, a, b, ops):
    ia, ib = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b)
    img = ia
    for i, op in enumerate(ops):
        img = kc.mix_process(img, ib if i % 2 == 0 else ia, op)
    return img.planes()


def expect(orc, a, b, ops):
    out = []
    for c in range(3):
        x = a[c]
        for i, op in enumerate(ops):
            x = orc.mix_plane(op.name if hasattr(op, "name") else op, x, (b if i % 2 == 0 else a)[c])
        out.append(x)
    return out + [np.ones_like(a[0])]


def cache_files(path):
    return sorted(f for f in os.listdir(path) if f.endswith(".kcco"))


def test_roundtrip_corruption_staleness_and_foreign_files(kc, tmp_path):
    from oracle import oracle as orc
    h, w = 48, 96
    a, b = synthetic_rgba(SEED_A, h, w), synthetic_rgba(SEED_B, h, w)
    ops_names = ["Add", "Multiply", "Subtract"]
    ops = [getattr(kc.MixType, n) for n in ops_names]
    want = expect(orc, a, b, ops_names)
    kc.set_specialize(2)  # compile at first sight: every evaluation below runs the compiled kernel or fails the test
    c0, s0 = kc.kernel_cache_stats(), kc.specialize_stats()
    assert_planes(program(kc, a, b, ops), want, what="compiled")
    c1, s1 = kc.kernel_cache_stats(), kc.specialize_stats()
    assert s1["kernels_compiled"] == s0["kernels_compiled"] + 1 and c1["files_written"] == c0["files_written"] + 1
    files = cache_files(tmp_path)
    assert len(files) == 1 and files[0].startswith("kc_chain_")
    path = os.path.join(tmp_path, files[0])
    good = open(path, "rb").read()
    assert good[:8] == b"KCCO0001"
    key_len, source_hash, code_len, code_hash = struct.unpack("<4Q", good[8:40])
    assert len(good) == 40 + key_len + code_len

    def next_process():
        kc.specialize_reset()
        return kc.kernel_cache_stats(), kc.specialize_stats()

    # ---- a later process: loaded, not compiled
    c0, s0 = next_process()
    assert_planes(program(kc, a, b, ops), want, what="loaded from the cache")
    c1, s1 = kc.kernel_cache_stats(), kc.specialize_stats()
    assert s1["kernels_compiled"] == s0["kernels_compiled"] and c1["files_accepted"] == c0["files_accepted"] + 1
    assert c1["kernels_loaded"] == c0["kernels_loaded"] + 1 and s1["specialized_launches"] > s0["specialized_launches"]

    def refused_then_recompiled(blob, what):
        with open(path, "wb") as f:
            f.write(blob)
        c0, s0 = next_process()
        assert_planes(program(kc, a, b, ops), want, what=what)
        c1, s1 = kc.kernel_cache_stats(), kc.specialize_stats()
        assert c1["files_refused"] == c0["files_refused"] + 1 and c1["files_accepted"] == c0["files_accepted"], what
        assert s1["kernels_compiled"] == s0["kernels_compiled"] + 1 and c1["files_written"] == c0["files_written"] + 1, what
        assert open(path, "rb").read() == good, what  # the fresh compile put the same code object back

    # ---- one flipped bit in the code
    flipped = bytearray(good)
    flipped[40 + key_len + code_len // 2] ^= 0x10
    refused_then_recompiled(bytes(flipped), "a corrupted code object")
    # ---- a truncated file, an empty one, garbage
    refused_then_recompiled(good[:len(good) // 2], "a truncated file")
    refused_then_recompiled(b"", "an empty file")
    refused_then_recompiled(os.urandom(4096), "garbage")
    # ---- made by another version of the generator / compiler: the stored source hash differs
    stale = bytearray(good)
    stale[16:24] = struct.pack("<Q", source_hash ^ 1)
    refused_then_recompiled(bytes(stale), "a stale file")
    # ---- trailing bytes
    refused_then_recompiled(good + b"\0", "a file with trailing bytes")
    # ---- another program's code object under this program's name (same length key): refused by the key comparison
    ops2_names = ["Multiply", "Add", "Subtract"]
    ops2 = [getattr(kc.MixType, n) for n in ops2_names]
    assert_planes(program(kc, a, b, ops2), expect(orc, a, b, ops2_names), what="second program")
    other = [f for f in cache_files(tmp_path) if f != files[0]]
    assert len(other) == 1
    refused_then_recompiled(open(os.path.join(tmp_path, other[0]), "rb").read(), "another program's file")


def test_first_evaluation_of_a_fresh_process_runs_the_cached_kernel(kc, tmp_path):
    """Mode 1 (the default: compile in the background after two sightings).  With the program's code object in the cache the FIRST
    sighting already launches the compiled kernel: no interpreter run, no compile."""
    from oracle import oracle as orc
    h, w = 40, 130
    a, b = synthetic_rgba(SEED_A, h, w), synthetic_rgba(SEED_B, h, w)
    names = ["Subtract", "Add", "Multiply", "Add"]
    ops = [getattr(kc.MixType, n) for n in names]
    want = expect(orc, a, b, names)
    kc.set_specialize(2)
    assert_planes(program(kc, a, b, ops), want, what="compile")
    kc.specialize_reset()
    kc.set_specialize(1, 2)
    s0 = kc.specialize_stats()
    assert_planes(program(kc, a, b, ops), want, what="first sighting in mode 1")
    s1 = kc.specialize_stats()
    assert s1["specialized_launches"] == s0["specialized_launches"] + 1 and s1["kernels_compiled"] == s0["kernels_compiled"]
    assert s1["compiles_pending"] == 0


def test_cache_off_compiles_every_time(kc, tmp_path):
    from oracle import oracle as orc
    h, w = 16, 64
    a, b = synthetic_rgba(SEED_A, h, w), synthetic_rgba(SEED_B, h, w)
    names = ["Add", "Subtract", "Multiply"]
    ops = [getattr(kc.MixType, n) for n in names]
    kc.kernel_cache_set_dir("off")
    kc.set_specialize(2)
    for _ in range(2):
        kc.specialize_reset()
        c0, s0 = kc.kernel_cache_stats(), kc.specialize_stats()
        assert_planes(program(kc, a, b, ops), expect(orc, a, b, names), what="cache off")
        c1, s1 = kc.kernel_cache_stats(), kc.specialize_stats()
        assert s1["kernels_compiled"] == s0["kernels_compiled"] + 1 and c1 == c0
    assert cache_files(tmp_path) == []
