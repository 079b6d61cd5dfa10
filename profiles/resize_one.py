#!/usr/bin/env python3
"""One resize case in a loop (for rocprofv3): python profiles/resize_one.py SRC DST [filter] [reps] [planes: 1 or 4]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kanter_core_amd as kc
from util import SEED_A, splitmix_plane

s, d = int(sys.argv[1]), int(sys.argv[2])
filt = kc.ResizeFilter.parse(sys.argv[3]) if len(sys.argv) > 3 else kc.ResizeFilter.Triangle
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
planes = int(sys.argv[5]) if len(sys.argv) > 5 else 1
kc.init(0)
kc.set_fusion(False)
src = kc.SlotImage.from_planes([splitmix_plane(SEED_A, c, s, s) for c in range(planes)])
for _ in range(reps):
    kc.resize_image(src, (d, d), filt)
kc.sync()
