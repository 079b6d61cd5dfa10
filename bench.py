#!/usr/bin/env python3
"""Headline benchmark: node-Mpix/s of the 32-node linear mix/invert graph on 4096x4096 f32x4
(BASELINE.json metric; SURVEY.md 8(d) config #3 at 4096^2) on one MI355X, plus the HBM roofline
fraction of the dominant kernel and the CPU oracle timed on this box's host cores.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ...] [--size 4096] [--nodes 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full evaluation of the graph (every node over every pixel) on inputs already
resident in HBM.  Rank 0 prints ONE JSON line.

Default workload:
    N = 1   chain32: the headline.  One launch per step (the 32 nodes are one fused chain); from the third
            evaluation on it is the run-time specialised kernel (csrc/specialize.cpp; the warm-up waits for the compile).
    N > 1   chain32_rows --size 8192: BASELINE config #3 -- the same 32-node graph on 8192x8192, every rank evaluating its
            row band of the result through the library's band path (kc_live_graph_evaluate_band, csrc/bands.cpp) and
            holding only those rows of the inputs.  A pointwise graph needs no halo and no exchange: "scaling": "strong"
            (fixed total work), time = max over ranks.

Other workloads (parity-tested configs of BASELINE.json, reported in DESIGN.md):
    --workload mix1           config #1: one Mix(Add) node, two 4096^2 f32x4 inputs
    --workload resize_blend   config #2: 512^2 -> 4096^2 Triangle resize + 3-node blend chain
    --workload chain32 --size 8192   config #3 at its full size (with N > 1: an independent graph per GPU, weak scaling)
    --workload chain32_rows --size 8192   config #3 split by row bands over the ranks through the library's band path
                              (kc_live_graph_evaluate_band; strong scaling, no exchange)
    --workload fanin          config #4 (any N, also N = 1): 8 independent 16-node subgraphs + a 7-node Mix(Add) tree as ONE
                              graph that the library's partitioner (kc_live_graph_partition) spreads over the ranks; branch
                              results go to the home rank as grouped RCCL send/recv of the planes, the join runs there;
                              per-rank host compute / exchange times in "per_rank"
Every workload run with N = 1 also carries a "parity" object: the timed graph's result against the oracle.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


def add_chain(kc, lg, src_a, src_b, n_nodes):
    """x0 = A; odd i: x_i = Mix(Add|Multiply)(x_{i-1}, B); even i: x_i = Mix(Subtract)(W, x_{i-1})
    with W = CombineRgba(Value 1.0 x3) (1x1, broadcast by the implicit resize).  Returns (first, last)."""
    one = lg.add_node(kc.Node.new(kc.NodeType.Value(1.0)))
    white = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
    for s in range(3):
        lg.connect(one, white, 0, s)
    prev, first = src_a, None
    for i in range(1, n_nodes + 1):
        if i & 1:
            n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply if (i >> 1) & 1 else kc.MixType.Add)))
            lg.connect(prev, n, 0, 0)
            lg.connect(src_b, n, 0, 1)
        else:
            n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
            lg.connect(white, n, 0, 0)
            lg.connect(prev, n, 0, 1)
        first = first if first is not None else n
        prev = n
    return first, prev


def embed(kc, lg, image, eid):
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, image), eid)
    return lg.add_node(kc.Node.new(kc.NodeType.Embed(eid)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=["chain32", "chain32_rows", "mix1", "resize_blend", "fanin"],
                    help="default: chain32 at 4096^2 (the headline) on one GPU; on several, chain32_rows at 8192^2 = BASELINE "
                         "config #3, the 32-node graph split by row bands through the library's band path")
    ap.add_argument("--policy", default="spread", choices=["spread", "auto"], help="fanin: placement policy of the partitioner")
    ap.add_argument("--size", type=int, default=None, help="default 4096; 8192 for the multi-GPU default workload")
    ap.add_argument("--nodes", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the unfused and PCIe-inclusive side measurements")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse the N > 1 control flow on one GPU")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "chain32" if args.gpus == 1 else "chain32_rows"
        if args.size is None and args.gpus > 1:
            args.size = 8192
    if args.size is None:
        args.size = 4096

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device_index = local_rank % max(torch.cuda.device_count(), 1)  # == local_rank on a full node
    torch.cuda.set_device(device_index)
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    import kanter_core_amd as kc
    from kanter_core_amd import multi_gpu
    from util import SEED_A, SEED_B, splitmix_plane, splitmix_rows

    kc.init(device_index)  # raises (no CPU fallback) when the HIP library or the GPU is missing
    # One explicit (non-default) HIP stream shared by torch (events, RCCL ordering) and the library.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    kc.set_stream(stream.cuda_stream)
    assert kc.get_stream() == stream.cuda_stream

    S, N = args.size, args.nodes
    tp = kc.TextureProcessor.new()

    def synth(seed, size, channels=4):
        return [splitmix_plane(seed + 0x100 * rank, c, size, size) for c in range(channels)]

    # ------------------------------------------------------------------ workloads
    # each returns: step(), node_px per step (this rank), algorithmic HBM bytes per step, description
    host_a = host_b = None
    rows = S
    band = None
    if args.workload == "chain32_rows":
        # strong scaling of ONE graph: this rank evaluates rows [y0, y1) of the result through the library's row-band
        # path (kc_live_graph_evaluate_band, csrc/bands.cpp) and holds only those rows of the inputs
        y0, y1 = multi_gpu.row_bands(S, world)[rank]
        rows = y1 - y0
        band = (y0, y1)
        full = lambda seed: [splitmix_rows(seed, c, S, S, y0, y1) for c in range(4)]  # noqa: E731
        host_a, host_b = full(SEED_A), full(SEED_B)
        args.workload = "chain32"
        band_note = " (row band %d:%d of %d, rank %d/%d)" % (y0, y1, S, rank, world)
    else:
        band_note = ""
    if args.workload == "chain32":
        if host_a is None:
            host_a, host_b = synth(SEED_A, S), synth(SEED_B, S)
        img_a, img_b = kc.SlotImage.from_planes(host_a), kc.SlotImage.from_planes(host_b)

        def make(use_cache):
            lg = tp.new_live_graph()
            lg.use_cache = use_cache
            if band is None:
                na, nb = embed(kc, lg, img_a, 0), embed(kc, lg, img_b, 1)
            else:
                lg.embed_slot_data_band(kc.SlotData(0, 0, img_a), 0, band[0], S)
                lg.embed_slot_data_band(kc.SlotData(0, 0, img_b), 1, band[0], S)
                na, nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(0))), lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
            first, last = add_chain(kc, lg, na, nb, N)
            return lg, na, first, last

        g = make(False)
        band_keep = []

        def make_cold(i):
            # an independent copy of the workload on its own inputs (see "cold" below)
            ia = kc.SlotImage.from_planes([splitmix_plane(SEED_A + 0x1000 * (i + 1), c, S, S) for c in range(4)])
            ib = kc.SlotImage.from_planes([splitmix_plane(SEED_B + 0x1000 * (i + 1), c, S, S) for c in range(4)])
            lgc = tp.new_live_graph()
            ca, cb = embed(kc, lgc, ia, 0), embed(kc, lgc, ib, 1)
            cfirst, clast = add_chain(kc, lgc, ca, cb, N)

            def cstep():
                lgc.connect(ca, cfirst, 0, 0)
                lgc.await_clean(clast)
            return cstep

        def step(gg=g):
            lg, na, first, last = gg
            if band is not None:
                band_keep[:] = [lg.evaluate_band(last, band[0], band[1])]  # stateless: every call evaluates the band
                return
            lg.connect(na, first, 0, 0)  # re-plugging the input dirties the whole chain (live_graph.rs:488-511)
            lg.await_clean(last)

        node_px = float(N) * S * rows
        # fused: R,G,B of A and B read once (24 B/px), R,G,B of the result written once (12 B/px);
        # alpha is a constant plane (0 B).  DESIGN.md "Kernels".
        alg_bytes = 36.0 * S * rows
        kernel = "chain_kernel<2,4,0>"
        desc = ("%d-node linear Mix graph (Add/Multiply alternating with invert = Mix(Subtract)(1, x)), "
                "%dx%d f32x4 per GPU, SURVEY 8(d) config #3%s" % (N, S, rows, band_note))
    elif args.workload == "mix1":
        host_a, host_b = synth(SEED_A, S), synth(SEED_B, S)
        img_a, img_b = kc.SlotImage.from_planes(host_a), kc.SlotImage.from_planes(host_b)
        lg = tp.new_live_graph()
        na, nb = embed(kc, lg, img_a, 0), embed(kc, lg, img_b, 1)
        m = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
        lg.connect(na, m, 0, 0)
        lg.connect(nb, m, 0, 1)
        g = (lg, na, m, m)

        def step():
            lg.connect(na, m, 0, 0)
            lg.await_clean(m)

        def make_cold(i):
            ia = kc.SlotImage.from_planes([splitmix_plane(SEED_A + 0x1000 * (i + 1), c, S, S) for c in range(4)])
            ib = kc.SlotImage.from_planes([splitmix_plane(SEED_B + 0x1000 * (i + 1), c, S, S) for c in range(4)])
            lgc = tp.new_live_graph()
            ca, cb = embed(kc, lgc, ia, 0), embed(kc, lgc, ib, 1)
            cm = lgc.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
            lgc.connect(ca, cm, 0, 0)
            lgc.connect(cb, cm, 0, 1)

            def cstep():
                lgc.connect(ca, cm, 0, 0)
                lgc.await_clean(cm)
            return cstep

        node_px, alg_bytes, kernel = float(S) * S, 36.0 * S * S, "chain_kernel<2,4,0>"
        desc = "single Mix(Add) node, two %dx%d f32x4 inputs, BASELINE config #1" % (S, S)
    elif args.workload == "resize_blend":
        s_small = S // 8
        host_a, host_b = synth(SEED_A, S), synth(SEED_B, s_small)
        img_a, img_b = kc.SlotImage.from_planes(host_a), kc.SlotImage.from_planes(host_b)
        lg = tp.new_live_graph()
        na, nb = embed(kc, lg, img_a, 0), embed(kc, lg, img_b, 1)
        n1 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
        n2 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply)))
        n3 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
        lg.connect(na, n1, 0, 0)
        lg.connect(nb, n1, 0, 1)
        lg.connect(n1, n2, 0, 0)
        lg.connect(na, n2, 0, 1)
        lg.connect(n2, n3, 0, 0)
        lg.connect(nb, n3, 0, 1)
        g = (lg, na, n1, n3)

        def step():
            lg.connect(na, n1, 0, 0)
            lg.await_clean(n3)

        def make_cold(i):
            ia = kc.SlotImage.from_planes([splitmix_plane(SEED_A + 0x1000 * (i + 1), c, S, S) for c in range(4)])
            ib = kc.SlotImage.from_planes([splitmix_plane(SEED_B + 0x1000 * (i + 1), c, s_small, s_small) for c in range(4)])
            lgc = tp.new_live_graph()
            ca, cb = embed(kc, lgc, ia, 0), embed(kc, lgc, ib, 1)
            c1 = lgc.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
            c2 = lgc.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply)))
            c3 = lgc.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
            for (a_, b_, s_) in ((ca, c1, 0), (cb, c1, 1), (c1, c2, 0), (ca, c2, 1), (c2, c3, 0), (cb, c3, 1)):
                lgc.connect(a_, b_, 0, s_)

            def cstep():
                lgc.connect(ca, c1, 0, 0)
                lgc.await_clean(c3)
            return cstep

        node_px = 3.0 * S * S
        # fused: the resampled B never exists in HBM.  One launch reads R,G,B of A (12 B/px) and of the
        # small B source, resamples B inside the kernel and writes R,G,B of the result (12 B/px).
        alg_bytes = (3 + 3) * 4.0 * S * S + 3 * 4.0 * s_small * s_small
        kernel = "resize_chain_kernel<2,3>"
        desc = "B %d^2 -> %d^2 Triangle resize + 3-node blend chain, BASELINE config #2" % (s_small, S)
    else:  # fanin
        # BASELINE config #4 as ONE graph that every rank builds: 8 independent 16-node subgraphs + a fixed-order 7-node
        # Mix(Add) tree.  The library's partitioner (csrc/partition.cpp) places the subgraphs on the ranks and the join
        # region on the home rank; multi_gpu.PartitionedEvaluator moves the cut slots (RCCL send / recv over xGMI, R, G, B
        # of each branch result; the constant alpha travels as a scalar).  Each rank embeds only the sources placed on it.
        from kanter_core_amd.multi_gpu import PartitionedEvaluator
        n_branches, sub_nodes = 8, 16
        lg = tp.new_live_graph()
        srcs, firsts, lasts = [], [], []
        for k in range(n_branches):
            na = lg.add_node(kc.Node.new(kc.NodeType.Embed(2 * k)))
            nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(2 * k + 1)))
            first, last = add_chain(kc, lg, na, nb, sub_nodes)
            srcs.append((na, nb))
            firsts.append(first)
            lasts.append(last)
        level = list(lasts)
        while len(level) > 1:
            nxt = []
            for i in range(0, len(level) - 1, 2):
                n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
                lg.connect(level[i], n, 0, 0)
                lg.connect(level[i + 1], n, 0, 1)
                nxt.append(n)
            if len(level) & 1:
                nxt.append(level[-1])
            level = nxt
        root = level[0]
        header_group = dist.new_group(backend="gloo") if (world > 1 and args.dist_backend == "nccl") else None
        ev = PartitionedEvaluator(lg, root, policy=kc.PartitionPolicy.Spread if args.policy == "spread" else kc.PartitionPolicy.Auto,
                                  device=torch.device("cuda", device_index), header_group=header_group)
        placed = {n: r for (n, r, _, _) in ev.plan.nodes}
        mine = [k for k in range(n_branches) if placed[lasts[k]] == rank]
        for k in range(n_branches):
            for j, (node, seed) in enumerate(zip(srcs[k], (0x5EED0100 + k, 0x5EED0200 + k))):
                if placed[node] == rank:
                    planes = [splitmix_plane(seed, c, S, S) for c in range(4)]
                    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(planes)), 2 * k + j)
        g = (lg, None, None, root)
        keep = []

        def step():
            for k in mine:
                lg.connect(srcs[k][0], firsts[k], 0, 0)  # re-plugging the input dirties the branch and what joins it
            keep[:] = [ev.evaluate()]

        n_tree = n_branches - 1
        node_px = float(len(mine) * sub_nodes + (n_tree if rank == ev.plan.home else 0)) * S * S
        # per rank: its fused subgraphs (36 B/px each); home: the add tree over 8 resident RGB results -- 4 launches reading
        # 2, 2, 3, 4 planes and writing 1 each, per channel.  An upper bound only: what the launches really move is decided
        # at run time (on one rank, pairs of branches and their Mix(Add) share a program: 5 launches, 25 plane passes per
        # channel) and is what the library counts -- the roofline below uses that count (algorithmic_bytes_per_step).
        alg_bytes = len(mine) * 36.0 * S * S + ((15 * 3 * 4.0 * S * S) if rank == ev.plan.home else 0.0)
        kernel = "chain_kernel<2,*> per subgraph + RCCL send/recv + chain_kernel<2..4,*> add tree"
        desc = ("8 independent 16-node subgraphs at %dx%d f32x4 placed by kc_live_graph_partition (%s), branch results sent to "
                "the home rank over RCCL, 7-node Mix(Add) tree there, BASELINE config #4" % (S, S, args.policy))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # A graph of the headline's shape on 1x1 constants: evaluating it runs the same host code as a real step (connect,
    # the walk, process_node, Mix, chain building) and launches nothing -- constants fold on the host.
    scratch = tp.new_live_graph()
    s_a = scratch.add_node(kc.Node.new(kc.NodeType.Value(0.25)))
    s_b = scratch.add_node(kc.Node.new(kc.NodeType.Value(0.5)))
    s_first, s_last = add_chain(kc, scratch, s_a, s_b, 32)

    def scratch_step():
        scratch.connect(s_a, s_first, 0, 0)
        scratch.await_clean(s_last)

    counted = [0.0]  # algorithmic bytes per step of the last timed() call, as the library counted them launch by launch
    host_us = []  # host time of every step of the last timed() call
    extra_warmup = [0]  # untimed steps run in addition to --warmup after a kernel compile landed (see warm())

    def warm(stepfn, warmup):
        # The first sightings of a chain program run through the interpreter while the program-specialised kernel
        # compiles on a worker thread (csrc/specialize.cpp); the rest of the warm-up starts once it has landed.
        # A graph's first evaluation builds plain chains and the programs that join chains / read more than four planes are
        # first seen at the second one (csrc/graph.cpp await_clean), so compiles are queued in up to three waves: pairs of
        # steps, each followed by a wait, until a pair queues nothing new.
        early = 0
        c0 = kc.specialize_stats()["kernels_compiled"]
        for _ in range(4):
            if early + 2 > warmup:
                break
            c1 = kc.specialize_stats()["kernels_compiled"]
            stepfn()
            stepfn()
            early += 2
            kc.specialize_wait()
            # (with several ranks every rank runs the same number of steps: a step of the fan-in workload exchanges planes)
            if world == 1 and early > 2 and kc.specialize_stats()["kernels_compiled"] == c1:
                break
        # The GPU sat idle while hiprtc ran (~2 s for the first program of a process) and its clocks went down with it:
        # when a compile did land, a few more untimed steps bring them back before the timed region (which stays EXACTLY
        # `steps` steps; the JSON reports the requested warm-up count and these extra ones separately).
        extra = 40 if (world > 1 or kc.specialize_stats()["kernels_compiled"] > c0) else 0
        extra_warmup[0] += extra
        for _ in range(max(warmup - early, 1 if early else 0) + extra):
            stepfn()

    def timed(stepfn, steps, warmup, sync_ranks=True):
        warm(stepfn, warmup)
        st0 = kc.stats()
        launches0 = st0["kernel_launches"]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # the host enqueues a step in about a third of the time the GPU needs for it, but the queue is empty right after the
        # synchronize below: a garbage-collection pause of the interpreter in the first steps shows up as GPU idle time
        # (a 20-step region is 2 ms).  Collect now, keep the collector out of the timed region.
        gc.collect()
        gc.disable()
        barrier() if sync_ranks else torch.cuda.synchronize()
        # The thread has just slept in a blocking synchronize (tens of ms after a long warm-up): its core is in a sleep state
        # and its caches are cold, and the first steps then take the host 170-430 us instead of 35 -- with an empty queue that is
        # GPU idle time (measured: host_us_per_step_first5).  A few ms of GPU-free host work wake it up before the region starts.
        spin_until = time.perf_counter() + 0.004
        while time.perf_counter() < spin_until:
            scratch_step()
        t0 = time.perf_counter()
        ev0.record(stream)
        host_us[:] = []
        for _ in range(steps):
            h0 = time.perf_counter()
            stepfn()
            host_us.append((time.perf_counter() - h0) * 1e6)
        ev1.record(stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gc.enable()
        barrier() if sync_ranks else torch.cuda.synchronize()
        st1 = kc.stats()
        counted[0] = (st1["algorithmic_bytes"] - st0["algorithmic_bytes"]) / float(steps)
        return t1 - t0, ev0.elapsed_time(ev1) * 1e-3, st1["kernel_launches"] - launches0

    def step_spread(stepfn, steps):
        """A second, separate pass with one HIP event after EVERY step (the events cost a few us of pipeline
        overlap between consecutive launches, so they stay out of the timed region above): sorted step times in us."""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        torch.cuda.synchronize()
        evs[0].record(stream)
        for i in range(steps):
            stepfn()
            evs[i + 1].record(stream)
        torch.cuda.synchronize()
        return sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(steps))

    wall, dev_s, launches = timed(step, args.steps, args.warmup)
    extra_main = extra_warmup[0]
    main_host_us = list(host_us)
    # Algorithmic bytes: the per-kernel figures of DESIGN.md section 3 summed by the library over the launches of a step
    # (kc_stats_algorithmic_bytes).  For the headline it must equal the closed form above (36 B/px); for graphs whose
    # fusion pattern is decided at run time (fanin: the add tree continues some of the branch chains) it is the figure.
    formula_bytes = alg_bytes
    alg_bytes = counted[0]
    if args.workload == "resize_blend" and kc.stats_counter("upsample_chain_launches"):
        # the integer-ratio up-sampling form (csrc/upsample_chain.inc) replaced resize_chain_kernel for this launch
        kernel = ("kc_upchain_<hash> (upsample_chain_tile<2,3,4,wide> with the chain program compiled to straight-line code at run "
                  "time, csrc/specialize.cpp; upsample_chain_kernel<2,3,true> = the same tile code driven by the step interpreter, "
                  "first sightings only)") if kc.specialize_stats()["specialized_launches"] else "upsample_chain_kernel<2,3,true>"
    elif kc.specialize_stats()["specialized_launches"] and "chain_kernel<" in kernel:
        kernel = "kc_chain_<hash> (the chain program compiled to straight-line code at run time, csrc/specialize.cpp; " + kernel + " = the interpreter, first sightings only)"
    main_step_us = step_spread(step, max(20, min(args.steps, 100)))
    total_px = node_px
    job_bytes, job_dev_s = alg_bytes, dev_s
    if world > 1:
        t = torch.tensor([wall, node_px, alg_bytes, dev_s], device=red_dev, dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        wall, total_px = float(tmax[0].item()), float(t[1].item())
        job_bytes, job_dev_s = float(t[2].item()), float(tmax[3].item())

    value = total_px * args.steps / wall / 1e6  # whole-job node-Mpix/s
    per_step_s = dev_s / args.steps             # HIP events on the launch stream over the timed region (this rank)
    launches_per_step = launches / args.steps
    # achieved = the job's algorithmic bytes / time / (n_gpu x 8 TB/s) -- all ranks' bytes over the slowest
    # rank's time (N = 1: this rank's)
    achieved = job_bytes / (job_dev_s / args.steps) / 1e9
    peak_gbs = HBM_PEAK_GBS * world
    # HBM traffic by PMC counters is collected in separate rocprofv3 --pmc passes (profiles/run_pmc.sh), never inside
    # this run: the figure below is read from the committed capture and labelled with its source; it is dropped when
    # the capture is of another kernel than the one this run launched.
    traffic = traffic_source = None
    pmc = os.path.join(ROOT, "profiles", "r03_pmc_upsample_chain_kernel.json" if args.workload == "resize_blend" else "r03_pmc_chain_kernel.json")
    if os.path.exists(pmc) and S == 4096 and ((args.workload == "chain32" and N == 32 and band is None) or args.workload == "resize_blend"):
        try:
            with open(pmc) as f:
                cap = json.load(f)
            ran_specialized = bool(kc.specialize_stats()["specialized_launches"])
            if ("kc_upchain_" in cap.get("kernel", "") and ran_specialized) if args.workload == "resize_blend" else (("kc_chain_" in cap.get("kernel", "")) == ran_specialized):
                traffic = cap.get("hbm_bytes_per_launch")
                traffic_source = "%s (kernel %s; %s)" % (os.path.relpath(pmc, ROOT), cap.get("kernel"), cap.get("captured") or "capture note missing")
        except Exception:
            traffic = traffic_source = None

    headline = args.workload == "chain32" and S == 4096 and N == 32 and not band_note
    out = {
        "metric": "node-Mpix/s on 4096x4096 f32x4, 32-node graph" if headline else "node-Mpix/s, %s" % args.workload,
        "value": round(value, 1),
        "unit": "Mpix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "extra_warmup_after_kernel_compile": extra_main,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if (args.workload == "fanin" or band_note) else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": desc, "graph_nodes": N if args.workload == "chain32" else None, "width": S, "height": S,
            "channels": 4, "use_cache": False,
            "series": ("%s at %dx%d: the line for --gpus N of this workload and size; N = 1 runs the same graph through the same path"
                       % ("chain32_rows (one graph split by row bands, strong scaling)" if band is not None else args.workload, S, S)),
            "cache_policy": kc.get_cache_policy(),
            "parallelism": (("row bands of one graph through kc_live_graph_evaluate_band, no exchange" if band is not None else
                             "independent graph per GPU") if args.workload != "fanin"
                            else "branches placed by the library's partitioner + RCCL send/recv to the home rank")
            if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": peak_gbs, "unit": "GB/s",
            "frac": round(achieved / peak_gbs, 4), "traffic": traffic, "traffic_source": traffic_source,
            "scope": "whole job: all ranks' algorithmic bytes / slowest rank's time / (n_gpus x 8 TB/s)" if world > 1 else "one GPU",
            "algorithmic_bytes_per_launch": alg_bytes / max(launches_per_step, 1.0) if launches_per_step else alg_bytes,
            "algorithmic_bytes_per_step": alg_bytes, "closed_form_bytes_per_step": formula_bytes,
            "kernel_us": round(per_step_s / max(launches_per_step, 1.0) * 1e6, 2),
            "launches_per_step": launches_per_step,
            # separate pass, one HIP event after every step (>= 20 steps): spread of individual steps, event overhead included
            "step_us_median": round(main_step_us[len(main_step_us) // 2], 2), "step_us_min": round(main_step_us[0], 2),
            "step_us_max": round(main_step_us[-1], 2),
            "specialized_kernel": bool(kc.specialize_stats()["specialized_launches"]),
            "host_us_per_step_first5": [round(x, 1) for x in main_host_us[:5]], "host_us_per_step_median": round(sorted(main_host_us)[len(main_host_us) // 2], 1),
        },
    }

    if rank == 0 and world == 1 and band is None and args.workload in ("chain32", "mix1", "resize_blend") and not args.no_extras:
        # ---- cold: nothing a step touches can still be in the 256 MB Infinity Cache ----
        # The timed region above re-evaluates ONE graph on the same inputs (what an editor does, and what the contract asks
        # for); with the cache policy (csrc/runtime.cpp, chain_cache_policy) one long-lived input of it stays in the Infinity
        # Cache from step to step.  Here four copies of the workload on different inputs are stepped in rotation, so every
        # step reads inputs and writes results that were last touched > 1.5 GB of traffic ago: HBM serves everything.
        cold_steps = [step] + [make_cold(i) for i in range(3)]
        rr = [0]

        def cold_step():
            cold_steps[rr[0] % len(cold_steps)]()
            rr[0] += 1

        kc_steps = max(20, min(args.steps, 100)) // len(cold_steps) * len(cold_steps)
        wall_c, dev_c, launches_c = timed(cold_step, kc_steps, 2 * len(cold_steps), sync_ranks=False)
        out["roofline"]["cold"] = {
            "kernel_us": round(dev_c / max(launches_c, 1) * 1e6, 2), "frac": round(counted[0] * kc_steps / dev_c / 1e9 / HBM_PEAK_GBS, 4),
            "steps": kc_steps,
            "how": "4 instances of the workload on different inputs evaluated in rotation: inputs and results of a step were last "
                   "touched more than 1.5 GB of traffic earlier, so the 256 MB Infinity Cache holds none of them",
        }
        out["roofline"]["warm_note"] = ("frac is the contract's measurement: one graph re-evaluated on resident inputs; with "
                                        "cache_policy = 1 a long-lived input can stay in the Infinity Cache between steps, which is how "
                                        "frac can exceed what HBM alone delivers (~0.80); cold.frac excludes that")
        del cold_steps

    if rank == 0 and args.workload == "chain32" and not args.no_extras and band is None:
        # ---- the same graph with every node materialised (use_cache = true): N launches per step ----
        gu = make(True)
        k2 = max(5, args.steps // 10)
        wall_u, dev_u, launches_u = timed(lambda: step(gu), k2, 2, sync_ranks=False)
        # unfused algorithmic bytes: two-plane Mix 36 B/px, invert (scalar left) 24 B/px
        unf_bytes = ((N + 1) // 2 * 36.0 + N // 2 * 24.0) * S * rows
        out["unfused"] = {
            "value": round(node_px * k2 / wall_u / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(wall_u / k2 * 1e3, 4),
            "launches_per_step": launches_u / k2, "achieved_GBps": round(unf_bytes * k2 / dev_u / 1e9, 1),
            "frac": round(unf_bytes * k2 / dev_u / 1e9 / HBM_PEAK_GBS, 4),
        }
        del gu
        # ---- PCIe-inclusive: host planes in, host planes out (never the headline value) ----
        t0 = time.perf_counter()
        ia, ib = kc.SlotImage.from_planes(host_a), kc.SlotImage.from_planes(host_b)
        lgp = tp.new_live_graph()
        pa, pb = embed(kc, lgp, ia, 0), embed(kc, lgp, ib, 1)
        _, lastp = add_chain(kc, lgp, pa, pb, N)
        res = lgp.await_clean(lastp).slot_data(lastp, 0).image.planes()
        pcie_s = time.perf_counter() - t0
        out["pcie_inclusive"] = {"value": round(node_px / pcie_s / 1e6, 1), "unit": "Mpix/s", "seconds": round(pcie_s, 4),
                                 "note": "8 pageable host planes uploaded, 4 downloaded, graph built and evaluated once"}
        del res, lgp

    if rank == 0 and world == 1 and args.workload == "chain32" and not args.no_cpu_baseline and band is None:
        from oracle import oracle as orc
        orc.set_threads(1)
        reps, cpu_s, ref = 0, 0.0, None
        while cpu_s < 10.0 and reps < 8:  # bounded sample: ~10-30 s of CPU work
            t0 = time.perf_counter()
            ref = orc.chain32(host_a, host_b, N)
            cpu_s += time.perf_counter() - t0
            reps += 1
        out["cpu_baseline"] = {
            "value": round(float(N) * S * S * reps / cpu_s / 1e6, 2), "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": "%d evaluations of the %d-node graph at %dx%d f32x4 (%.1f s total), node by node, planes "
                      "sequentially, as the reference's one-thread-per-node engine runs a linear chain" % (reps, N, S, S, cpu_s),
            "host_cores_available": os.cpu_count(),
        }
        # the same loops, rows split over the host cores of this GPU's share of the box (SURVEY 8(d) ii)
        threads = max(1, min(os.cpu_count() or 1, 16))
        if threads > 1:
            orc.set_threads(threads)
            orc.chain32(host_a, host_b, N)  # first touch / thread start-up
            t0 = time.perf_counter()
            for _ in range(3):
                orc.chain32(host_a, host_b, N)
            mt_s = time.perf_counter() - t0
            orc.set_threads(1)
            out["cpu_baseline_all_cores"] = {"value": round(float(N) * S * S * 3 / mt_s / 1e6, 2), "unit": "Mpix/s",
                                             "cores": threads, "kind": "port",
                                             "note": "a reported baseline, not a target: %d of the box's %d cores (one GPU's share); the port keeps the "
                                                     "reference's per-node plane allocation and first touch, which do not parallelise" % (threads, os.cpu_count() or 1),
                                             "sample": "3 evaluations, rows split over %d OpenMP threads = one GPU's share of the box's %d host cores (%.1f s)" % (threads, os.cpu_count() or 1, mt_s)}
        # parity of the timed workload against the oracle, on the same inputs
        got = g[0].slot_data(g[3], 0).image.planes()
        mism = int(sum((x.view(np.uint32) != y.view(np.uint32)).sum() for x, y in zip(got, ref)))
        out["parity"] = {"checked_pixels": S * rows * 4, "bit_mismatches": mism}

    if rank == 0 and world == 1 and args.workload in ("mix1", "resize_blend", "fanin") and not args.no_cpu_baseline:
        # parity of the timed graph against the oracle on the same inputs (the oracle as checker, after the timed region)
        from oracle import oracle as orc
        orc.set_threads(max(1, min(os.cpu_count() or 1, 16)))
        if args.workload == "mix1":
            ref = [orc.mix_plane("Add", host_a[c], host_b[c]) for c in range(3)]
        elif args.workload == "resize_blend":
            bu = [orc.resize_plane(p, S, S, "Triangle") for p in host_b[:3]]
            ref = [orc.mix_plane("Subtract", orc.mix_plane("Multiply", orc.mix_plane("Add", host_a[c], bu[c]), host_a[c]), bu[c])
                   for c in range(3)]
        else:
            parts = []
            for k in range(n_branches):
                ha = [splitmix_plane(0x5EED0100 + k, c, S, S) for c in range(3)]
                hb = [splitmix_plane(0x5EED0200 + k, c, S, S) for c in range(3)]
                parts.append(orc.chain32(ha, hb, sub_nodes)[:3])
            while len(parts) > 1:
                nxt = [[orc.mix_plane("Add", parts[i][c], parts[i + 1][c]) for c in range(3)] for i in range(0, len(parts) - 1, 2)]
                if len(parts) & 1:
                    nxt.append(parts[-1])
                parts = nxt
            ref = parts[0]
        orc.set_threads(1)
        ref = list(ref) + [np.ones((S, S), np.float32)]
        got = g[0].slot_data(g[3], 0).image.planes()
        nan_ok = lambda x, y: (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))  # noqa: E731
        mism = int(sum((~nan_ok(x, y)).sum() for x, y in zip(got, ref)))
        out["parity"] = {"checked_pixels": S * S * 4, "bit_mismatches": mism}

    if band is not None and not args.no_cpu_baseline:
        # parity of the row-band path on every rank: three 32-row crops of this rank's band (its first, middle and last rows)
        # against the oracle on the same rows of the inputs -- the graph is pointwise, so a crop's result depends on nothing else
        from oracle import oracle as orc
        got = band_keep[0].planes()
        mism, checked = 0, 0
        for r0 in sorted({0, max(0, rows // 2 - 16), max(0, rows - 32)}):
            r1 = min(rows, r0 + 32)
            ref = orc.chain32([p[r0:r1] for p in host_a], [p[r0:r1] for p in host_b], N)
            mism += int(sum((x[r0:r1].view(np.uint32) != np.ascontiguousarray(y).view(np.uint32)).sum() for x, y in zip(got, ref)))
            checked += (r1 - r0) * S * 4
        if world > 1:
            t = torch.tensor([mism, checked], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            mism, checked = int(t[0].item()), int(t[1].item())
        out["parity"] = {"checked_pixels": checked, "bit_mismatches": mism,
                         "how": "3 crops of 32 rows per rank against the oracle on the same rows, summed over the ranks"}

    if args.workload == "fanin" and world > 1 and not args.no_cpu_baseline:
        # parity of the partitioned evaluation: the home rank's result against the oracle on three 32-row crops (the graph
        # is pointwise); every rank learns the verdict
        verdict = torch.zeros(2, device=red_dev, dtype=torch.float64)
        if rank == ev.plan.home:
            from oracle import oracle as orc
            got = keep[0].planes()
            for r0 in (0, S // 2 - 16, S - 32):
                parts = []
                for k in range(n_branches):
                    ha = [splitmix_rows(0x5EED0100 + k, c, S, S, r0, r0 + 32) for c in range(3)]
                    hb = [splitmix_rows(0x5EED0200 + k, c, S, S, r0, r0 + 32) for c in range(3)]
                    parts.append(orc.chain32(ha, hb, sub_nodes)[:3])
                while len(parts) > 1:
                    nxt = [[orc.mix_plane("Add", parts[i][c], parts[i + 1][c]) for c in range(3)] for i in range(0, len(parts) - 1, 2)]
                    if len(parts) & 1:
                        nxt.append(parts[-1])
                    parts = nxt
                verdict[0] += int(sum((got[c][r0:r0 + 32].view(np.uint32) != np.ascontiguousarray(parts[0][c]).view(np.uint32)).sum() for c in range(3)))
                verdict[1] += 32 * S * 3
        dist.all_reduce(verdict, op=dist.ReduceOp.SUM)
        out["parity"] = {"checked_pixels": int(verdict[1].item()), "bit_mismatches": int(verdict[0].item()),
                         "how": "the home rank's result, 3 crops of 32 rows, against the oracle"}

    if args.workload == "fanin":
        st = dict(ev.stats)
        st["transfers_in_plan"] = len(ev.plan.transfers)
        st["branches"] = len(mine)
        if world > 1:
            allst = [None] * world
            dist.all_gather_object(allst, st, group=header_group)
        else:
            allst = [st]
        out["per_rank"] = [{k: (round(v, 6) if isinstance(v, float) else v) for k, v in x.items()} for x in allst]
        out["multi_gpu_measured"] = world > 1 and args.dist_backend == "nccl"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
