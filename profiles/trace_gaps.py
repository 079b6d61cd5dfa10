#!/usr/bin/env python3
"""Kernel durations and the gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV (last N dispatches).
    python profiles/trace_gaps.py <dir with *_kernel_trace.csv> [N]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 140
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    out.append((r["Kernel_Name"][:24], (e - s) / 1e3, gap))
    prev_end = e
for i in range(0, len(out), 10):
    print(" ".join("%s:%.0f/%+.0f" % (k[-8:], d, g) for k, d, g in out[i:i + 10]))
