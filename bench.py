#!/usr/bin/env python3
"""Headline benchmark: node-Mpix/s of the 32-node linear mix/invert graph on 4096x4096 f32x4
(BASELINE.json metric; SURVEY.md 8(d) config #3 at 4096^2), per MI355X, plus the HBM roofline
fraction of the dominant kernel and the CPU oracle timed on this box's host cores.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4096] [--nodes 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full evaluation of the graph (every one of its 32 Mix nodes over every pixel) on
inputs already resident in HBM.  With N > 1 every rank evaluates its own graph on its own GPU
(independent graphs: no data-path collective, weak scaling); the time is the max over ranks.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


def build_chain(kc, lg, img_a, img_b, n_nodes):
    """x0 = A; odd i: x_i = Mix(Add|Multiply)(x_{i-1}, B); even i: x_i = Mix(Subtract)(W, x_{i-1})
    with W = CombineRgba(Value 1.0 x3) (1x1, broadcast by the implicit resize)."""
    na = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
    nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, img_a), 0)
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, img_b), 1)
    one = lg.add_node(kc.Node.new(kc.NodeType.Value(1.0)))
    white = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
    for s in range(3):
        lg.connect(one, white, 0, s)
    prev, first = na, None
    for i in range(1, n_nodes + 1):
        if i & 1:
            n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply if (i >> 1) & 1 else kc.MixType.Add)))
            lg.connect(prev, n, 0, 0)
            lg.connect(nb, n, 0, 1)
        else:
            n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
            lg.connect(white, n, 0, 0)
            lg.connect(prev, n, 0, 1)
        first = first if first is not None else n
        prev = n
    return na, first, prev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-size", type=int, default=0, help="plane size of the CPU baseline sample (0 = --size)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    torch.cuda.set_device(local_rank)

    import kanter_core_amd as kc
    from util import SEED_A, SEED_B, splitmix_plane

    kc.init(local_rank)  # raises (no CPU fallback) when the HIP library or the GPU is missing
    # One explicit (non-default) HIP stream shared by torch (events, RCCL ordering) and the library.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    kc.set_stream(stream.cuda_stream)
    assert kc.get_stream() == stream.cuda_stream

    S, N = args.size, args.nodes
    a = [splitmix_plane(SEED_A + 0x100 * rank, c, S, S) for c in range(4)]
    b = [splitmix_plane(SEED_B + 0x100 * rank, c, S, S) for c in range(4)]
    img_a, img_b = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b)

    tp = kc.TextureProcessor.new()

    def make_graph(use_cache):
        lg = tp.new_live_graph()
        lg.use_cache = use_cache
        return (lg,) + build_chain(kc, lg, img_a, img_b, N)

    def step(g):
        lg, na, first, last = g
        lg.connect(na, first, 0, 0)  # re-plugging the input dirties the whole chain (live_graph.rs:488-511)
        lg.await_clean(last)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(g, steps, warmup):
        for _ in range(warmup):
            step(g)
        launches0 = kc.stats()["kernel_launches"]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(steps):
            step(g)
        ev1.record(stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        launches = kc.stats()["kernel_launches"] - launches0
        return t1 - t0, ev0.elapsed_time(ev1) * 1e-3, launches

    g = make_graph(False)
    wall, dev_s, launches = timed(g, args.steps, args.warmup)
    if world > 1:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    node_px = float(N) * S * S
    value = world * node_px * args.steps / wall / 1e6  # whole-job node-Mpix/s

    # ---- roofline of the dominant kernel (the fused chain kernel: one launch per step) -----
    # Algorithmic bytes per launch: R,G,B of A and of B read once (24 B/px), R,G,B of the result
    # written once (12 B/px); alpha is a constant plane (0 B).  See DESIGN.md "Kernels".
    bytes_per_launch = 36.0 * S * S
    kernel_s = dev_s / max(launches, 1)  # HIP events on the launch stream over the timed region
    achieved = bytes_per_launch / kernel_s / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_chain_kernel.json")
    if os.path.exists(pmc) and S == 4096 and N == 32:
        try:
            with open(pmc) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "node-Mpix/s on 4096x4096 f32x4, 32-node graph",
        "value": round(value, 1),
        "unit": "Mpix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "32-node linear Mix graph (Add/Multiply alternating with invert = Mix(Subtract)(1, x)), "
                        "%dx%d f32x4 per GPU, SURVEY 8(d) config #3" % (S, S),
            "graph_nodes": N, "width": S, "height": S, "channels": 4, "use_cache": False,
            "parallelism": "independent graph per GPU" if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm", "kernel": "chain_kernel<2,2,false>", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "algorithmic_bytes_per_launch": bytes_per_launch,
            "kernel_us": round(kernel_s * 1e6, 2), "launches_per_step": launches / args.steps,
        },
    }

    if rank == 0:
        # ---- the same graph with every node materialised (use_cache = true): 32 launches/step ----
        gu = make_graph(True)
        k2 = max(5, args.steps // 10)
        wall_u, dev_u, launches_u = timed_local(gu, k2, 2, step, kc, torch, stream)
        # unfused algorithmic bytes: 16 x 36 B/px (two-plane Mix) + 16 x 24 B/px (invert, scalar left)
        unf_bytes = (16 * 36.0 + 16 * 24.0) * S * S * (N / 32.0)
        out["unfused"] = {
            "value": round(node_px * k2 / wall_u / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(wall_u / k2 * 1e3, 4),
            "launches_per_step": launches_u / k2, "achieved_GBps": round(unf_bytes * k2 / dev_u / 1e9, 1),
            "frac": round(unf_bytes * k2 / dev_u / 1e9 / HBM_PEAK_GBS, 4),
        }
        del gu

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        cs = args.cpu_size or S
        ca = a if cs == S else [p[:cs, :cs].copy() for p in a]
        cb = b if cs == S else [p[:cs, :cs].copy() for p in b]
        orc.set_threads(1)
        t0 = time.perf_counter()
        ref = orc.chain32(ca, cb, N)
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": round(float(N) * cs * cs / cpu_s / 1e6, 2), "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": "1 evaluation of the %d-node graph at %dx%d f32x4 (%.1f s), node by node, planes "
                      "sequentially, as the reference's one-thread-per-node engine runs a linear chain" % (N, cs, cs, cpu_s),
            "host_cores_available": os.cpu_count(),
        }
        # parity of the timed workload against the oracle, on the same inputs
        got = g[0].slot_data(g[3], 0).image.planes()
        sub = [p[:cs, :cs] for p in got]
        mism = int(sum((x.view(np.uint32) != y.view(np.uint32)).sum() for x, y in zip(sub, ref)))
        out["parity"] = {"checked_pixels": cs * cs * 4, "bit_mismatches": mism}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def timed_local(g, steps, warmup, step, kc, torch, stream):
    for _ in range(warmup):
        step(g)
    launches0 = kc.stats()["kernel_launches"]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(steps):
        step(g)
    ev1.record(stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return t1 - t0, ev0.elapsed_time(ev1) * 1e-3, kc.stats()["kernel_launches"] - launches0


if __name__ == "__main__":
    main()
