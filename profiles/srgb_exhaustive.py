#!/usr/bin/env python3
"""Every float in [0, 2) -- all 1 065 353 217 bit patterns of [0, 1] and the clamped range above it -- through the device's
sRGB export (to_u8_kernel<true>: hardware log2 / exp2 estimate + ONE table comparison each way) against the threshold
table's definition q(x) = #{v : x >= T[v]} - 1.  The table itself is checked against the direct formula with libm's powf for
every float in [0, 1] by tools/gen_srgb_thresholds.c.     python profiles/srgb_exhaustive.py   (~1 minute on a GPU box)"""
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import kanter_core_amd as kc

text = open(os.path.join(ROOT, "kanter_core_amd", "csrc", "srgb_thresholds.inc")).read()
T = np.array([int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{8})u", text)], dtype=np.uint32)
assert len(T) == 256 and T[0] == 0
kc.init(0)
S = 4096
bad = total = 0
t0 = time.time()
for chunk in range(64):
    bits = (np.arange(S * S, dtype=np.uint32) + np.uint32(chunk * S * S)).reshape(S, S)
    x = bits.view(np.float32)
    got = kc.SlotImage.from_planes([x]).to_u8(True)[:, :, 0]
    want = (np.searchsorted(T, bits.reshape(-1), side="right") - 1).astype(np.uint8).reshape(S, S)
    n = int((got != want).sum())
    if n:
        i = np.argwhere(got != want)[0]
        print("chunk %d: %d mismatches, first at bits 0x%08x: got %d want %d" % (chunk, n, bits[tuple(i)], got[tuple(i)], want[tuple(i)]))
    bad += n
    total += S * S
print("%d floats (bit patterns 0 .. 0x%08x) through to_u8_kernel<srgb>: %d mismatches against the table's definition (%.0f s)"
      % (total, total - 1, bad, time.time() - t0))
sys.exit(1 if bad else 0)
