#!/usr/bin/env python3
"""Per-kernel HBM bandwidth on one MI355X (4096x4096 planes): every kernel of csrc/kernels.hip
through the C-ABI operator entry points, HIP-event timed on the shared stream.
    python profiles/kernel_microbench.py [--size 4096] [--reps 50]
Prints one JSON object: kernel -> {us, GB/s, frac of 8 TB/s, algorithmic bytes}."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import numpy as np
    import torch
    import kanter_core_amd as kc
    from util import SEED_A, SEED_B, splitmix_plane

    torch.cuda.set_device(0)
    kc.init(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    kc.set_stream(stream.cuda_stream)
    S = args.size
    px = float(S) * S
    a = [splitmix_plane(SEED_A, c, S, S) for c in range(4)]
    b = [splitmix_plane(SEED_B, c, S, S) for c in range(4)]
    A, B = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b)
    Ag, Bg = kc.SlotImage.from_planes(a[:1]), kc.SlotImage.from_planes(b[:1])
    small = kc.SlotImage.from_planes([splitmix_plane(SEED_B, 0, S // 8, S // 8)])
    small4 = kc.SlotImage.from_planes([splitmix_plane(SEED_B, c, S // 8, S // 8) for c in range(4)])
    white = kc.combine_rgba_process([kc.value_process(1.0)] * 3 + [None])
    u8 = np.random.default_rng(1).integers(0, 256, (S, S, 4), dtype=np.uint8)
    kc.set_fusion(False)  # every operator call launches its kernel immediately

    def timed(fn, reps=args.reps):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    cases = {
        "mix_add_rgba (chain_kernel<2,4,0>)": (lambda: kc.mix_process(A, B, kc.MixType.Add), 36 * px),
        "mix_add_gray (chain_kernel<2,4,0>)": (lambda: kc.mix_process(Ag, Bg, kc.MixType.Add), 12 * px),
        "invert_rgba (chain_kernel<1,4,0>)": (lambda: kc.mix_process(kc.resize_image(white, (S, S)), A, kc.MixType.Subtract), 24 * px),
        "mix_divide_rgba (chain_kernel<2,4,1>)": (lambda: kc.mix_process(A, B, kc.MixType.Divide), 36 * px),
        "mix_pow_rgba (chain_kernel<2,1,2>)": (lambda: kc.mix_process(A, B, kc.MixType.Pow), 36 * px),
        "as_type rgba->gray (chain_kernel<3,4,1>)": (lambda: A.as_type(False), 16 * px),
        "fill (fill_kernel)": (lambda: kc.SlotImage.from_value((S, S), 0.5, False).materialize(), 4 * px),
        "resize 512->4096 triangle, 1 plane (resize_lds_kernel<2,3>)": (lambda: kc.resize_image(small, (S, S)), 4 * px * (1 + 1 / 64.0)),
        "resize 512->4096 triangle, 4 planes in one launch": (lambda: kc.resize_image(small4, (S, S)), 16 * px * (1 + 1 / 64.0)),
        "resize 4096->512 triangle, 1 plane": (lambda: kc.resize_image(Ag, (S // 8, S // 8)), 4 * px * (1 + 1 / 64.0)),
        "resize 512->4096 lanczos3, 1 plane": (lambda: kc.resize_image(small, (S, S), kc.ResizeFilter.Lanczos3), 4 * px * (1 + 1 / 64.0)),
        "height_to_normal (height_to_normal_kernel)": (lambda: kc.height_to_normal_process(Ag), 16 * px),
    }
    out = {}
    for name, (fn, nbytes) in cases.items():
        t = timed(fn, 5 if "pow" in name else args.reps)
        out[name] = {"us": round(t * 1e6, 1), "GBps": round(nbytes / t / 1e9, 1), "frac": round(nbytes / t / 8e12, 3),
                     "algorithmic_bytes": nbytes}
    # u8 boundary: kernel + PCIe copy are inseparable through the ABI; report end to end
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        A.to_u8()
    out["to_u8 rgba incl. D2H copy"] = {"us": round((time.perf_counter() - t0) / 5 * 1e6, 1)}
    t0 = time.perf_counter()
    for _ in range(5):
        kc.SlotImage.from_u8(u8)
    out["from_u8 rgba incl. H2D copy"] = {"us": round((time.perf_counter() - t0) / 5 * 1e6, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
