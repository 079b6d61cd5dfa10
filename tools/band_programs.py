#!/usr/bin/env python3
"""Evaluates the row bands a 2 / 4 / 8-rank run of BASELINE configs #3 (8192^2) and #4 (4096^2) cuts, one band each, so that their
chain programs are compiled and recorded (tools/dump_baseline_programs.sh).  The programs depend on the band's shape only through
the cache-policy mask (bytes streamed against the Infinity Cache), so one band per world size is enough."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import kanter_core_amd as kc  # noqa: E402
from bench import add_chain  # noqa: E402
from util import splitmix_rows  # noqa: E402

kc.init(0)
tp = kc.TextureProcessor.new()


def fanin(lg):
    lasts = []
    for k in range(8):
        na = lg.add_node(kc.Node.new(kc.NodeType.Embed(2 * k)))
        nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(2 * k + 1)))
        lasts.append(add_chain(kc, lg, na, nb, 16)[1])
    while len(lasts) > 1:
        nxt = []
        for i in range(0, len(lasts) - 1, 2):
            n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
            lg.connect(lasts[i], n, 0, 0)
            lg.connect(lasts[i + 1], n, 0, 1)
            nxt.append(n)
        lasts = nxt
    return lasts[0]


for world in (2, 4, 8):
    for (S, which) in ((8192, "chain"), (4096, "fanin")):
        rows = S // world
        lg = tp.new_live_graph()
        n_src = 2 if which == "chain" else 16
        for e in range(n_src):
            planes = [splitmix_rows(0x5EED0100 + e, c, S, S, 0, rows) for c in range(4)]
            lg.embed_slot_data_band(kc.SlotData(0, 0, kc.SlotImage.from_planes(planes)), e, 0, S)
        if which == "chain":
            na, nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(0))), lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
            root = add_chain(kc, lg, na, nb, 32)[1]
        else:
            root = fanin(lg)
        for _ in range(6):
            img = lg.evaluate_band(root, 0, rows)
            kc.specialize_wait()
        del img, lg
        kc.sync()
        kc.pool_trim()
print("band programs done", kc.specialize_stats())
