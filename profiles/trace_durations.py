#!/usr/bin/env python3
"""Percentiles of per-dispatch durations from a rocprofv3 kernel-trace CSV, for kernels whose name contains a pattern.
    python3 profiles/trace_durations.py <dir with *_kernel_trace.csv> <pattern> [--seq]"""
import csv
import glob
import os
import sys

d, pat = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
for f in files:
    rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    if not rows:
        continue
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    s = sorted(dur)
    q = lambda p: s[min(len(s) - 1, int(p * len(s)))]
    print("%s: n=%d min=%.1f p10=%.1f p50=%.1f p90=%.1f max=%.1f mean=%.2f" % (rows[0]["Kernel_Name"][:60], len(s), s[0], q(.1), q(.5), q(.9), s[-1], sum(s) / len(s)))
    gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows) - 1)]
    if gaps:
        g = sorted(gaps)
        print("  gaps between consecutive dispatches: p10=%.1f p50=%.1f p90=%.1f" % (g[int(.1 * len(g))], g[len(g) // 2], g[int(.9 * len(g))]))
    if "--seq" in sys.argv:
        print("  " + " ".join("%.0f" % x for x in dur))
