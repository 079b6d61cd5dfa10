#!/usr/bin/env python3
"""HeightToNormal on square planes of several sizes in a loop (for rocprofv3): python profiles/h2n_sizes.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kanter_core_amd as kc
from util import SEED_A, splitmix_plane

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kc.init(0)
for s in (1024, 2048, 3000, 4096, 8192):
    img = kc.SlotImage.from_planes([splitmix_plane(SEED_A, 0, s, s)])
    for _ in range(reps):
        kc.height_to_normal_process(img)
    kc.sync()
