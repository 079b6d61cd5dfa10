#!/usr/bin/env python3
"""Tile-shape sweep for the LDS resize kernel on one MI355X (tuning aid, not part of the product).
    python profiles/resize_tile_sweep.py [--reps 100]
For every (source, destination, filter) case: the default tile choice, the two-pass form, and every
override the kernel accepts (KC_RESIZE_TILE_W/H are read by kc_init, so each setting re-initialises
the library).  Prints one line per setting: microseconds per plane."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--cases", default="all")
    args = ap.parse_args()
    import torch
    import kanter_core_amd as kc
    from util import SEED_A, splitmix_plane

    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)

    def setup(env):
        if kc.is_initialized():
            kc.shutdown()
        for k in ("KC_RESIZE_MODE", "KC_RESIZE_TILE_W", "KC_RESIZE_TILE_H"):
            os.environ[k] = str(env.get(k, 0))
        kc.init(0)
        kc.set_stream(stream.cuda_stream)
        kc.set_fusion(False)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(args.reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.reps

    F = kc.ResizeFilter
    cases = [
        ("up 512->4096 tri", 512, 4096, F.Triangle),
        ("up 512->4096 lanczos3", 512, 4096, F.Lanczos3),
        ("up 2048->4096 tri", 2048, 4096, F.Triangle),
        ("up 1000->4096 catmull", 1000, 4096, F.CatmullRom),
        ("down 4096->512 tri", 4096, 512, F.Triangle),
        ("down 4096->2048 tri", 4096, 2048, F.Triangle),
        ("down 4096->1024 lanczos3", 4096, 1024, F.Lanczos3),
        ("down 4096->3000 tri", 4096, 3000, F.Triangle),
    ]
    tiles = [(1024, 16), (1024, 8), (512, 16), (512, 32), (512, 8), (256, 64), (256, 32), (256, 16), (256, 8), (128, 32),
             (128, 16), (128, 64), (64, 64), (64, 32), (64, 16), (64, 8), (32, 32), (32, 16), (32, 8), (16, 16), (16, 8), (16, 4),
             (64, 4), (32, 4), (128, 4), (128, 8), (8, 4), (8, 8), (256, 4)]
    planes = {}
    for name, s, d, filt in cases:
        if args.cases != "all" and args.cases not in name:
            continue
        res = []
        for label, env in [("default", {}), ("two-pass", {"KC_RESIZE_MODE": 3})] + [
                ("%dx%d" % t, {"KC_RESIZE_TILE_W": t[0], "KC_RESIZE_TILE_H": t[1]}) for t in tiles]:
            setup(env)
            if s not in planes:
                planes[s] = splitmix_plane(SEED_A, 0, s, s)
            src = kc.SlotImage.from_planes([planes[s]])
            st0 = kc.stats()
            us = timed(lambda: kc.resize_image(src, (d, d), filt))
            res.append((us, label))
            del src
        # an override the kernel cannot take falls back to the default choice: same time as "default"
        res_sorted = sorted(res[2:])[:6]
        print("%-26s default %7.1f  two-pass %7.1f | best: %s" % (
            name, res[0][0], res[1][0], "  ".join("%s %.1f" % (l, u) for u, l in res_sorted)), flush=True)


if __name__ == "__main__":
    main()
