#!/usr/bin/env python3
"""The streaming kernels that are not the chain kernel, each in a loop (for rocprofv3 --kernel-trace --stats):
to_u8, from_u8, height_to_normal, Mix(Pow), Mix(Divide), fill, RGBA->gray.   python profiles/stream_kernels.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import kanter_core_amd as kc
from util import SEED_A, SEED_B, splitmix_plane

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S = 4096
kc.init(0)
kc.set_fusion(False)
kc.set_specialize(int(os.environ.get("KC_SPECIALIZE", "2")))
a = [splitmix_plane(SEED_A, c, S, S) for c in range(4)]
b = [splitmix_plane(SEED_B, c, S, S) for c in range(4)]
A, B = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b)
Ag = kc.SlotImage.from_planes(a[:1])
u8 = np.random.default_rng(1).integers(0, 256, (S, S, 4), dtype=np.uint8)
for _ in range(reps):
    A.to_u8()
for _ in range(max(reps // 4, 2)):
    A.to_u8(True)
for _ in range(reps):
    kc.SlotImage.from_u8(u8)
for _ in range(reps):
    kc.height_to_normal_process(Ag)
for _ in range(reps):
    kc.mix_process(A, B, kc.MixType.Pow)
for _ in range(reps):
    kc.mix_process(A, B, kc.MixType.Divide)
for _ in range(reps):
    kc.mix_process(A, B, kc.MixType.Add)
for _ in range(reps):
    A.as_type(False)
for _ in range(reps):
    kc.SlotImage.from_value((S, S), 0.5, False).materialize()
kc.sync()
