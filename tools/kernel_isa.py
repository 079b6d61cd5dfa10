#!/usr/bin/env python3
"""Instruction histogram / register use of one kernel in an AMDGPU assembly listing.
    hipcc --offload-arch=gfx950 ... -S --cuda-device-only -o k.s kernels.hip;  python tools/kernel_isa.py k.s height_to_normal_kernelILb0"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 18
i = 0
while i < len(text):
    m = re.match(r"^(\S+):\s", text[i] + " ")
    if m and pat in m.group(1) and not m.group(1).startswith("."):
        name = m.group(1)
        hist = collections.Counter()
        j = i + 1
        while j < len(text) and "s_endpgm" not in text[j]:
            t = text[j].strip()
            if t and re.match(r"^[a-z]", t) and not t.endswith(":"):
                hist[t.split()[0]] += 1
            j += 1
        print(name, "instructions:", sum(hist.values()), " VALU:", sum(v for k, v in hist.items() if k.startswith("v_")),
              " SALU:", sum(v for k, v in hist.items() if k.startswith("s_")))
        print("  " + ", ".join("%s %d" % kv for kv in hist.most_common(top)))
        i = j
    i += 1
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n){0,12}?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n){0,6}?\s+\.vgpr_count:\s+(\d+)", "\n".join(text)):
    if pat in m.group(1):
        print("  sgpr", m.group(2), "vgpr", m.group(3))
