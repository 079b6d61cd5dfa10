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
    N = 1   chain32: the headline.  One launch per step (the 32 nodes are one fused chain), the kernel compiled for that
            program (csrc/specialize.cpp) -- taken from the kernel cache at the first evaluation (the build pre-compiles the
            BASELINE programs), so the warm-up is exactly --warmup steps.
    N > 1   chain32_rows --size 8192: BASELINE config #3 -- the same 32-node graph on 8192x8192, every rank evaluating its
            row band of the result through the library's band path (kc_live_graph_evaluate_band, csrc/bands.cpp) and
            holding only those rows of the inputs.  A pointwise graph needs no halo and no exchange: "scaling": "strong"
            (fixed total work), time = max over ranks.  A side leg outside the timed region ("gather_to_rank0") then moves the
            finished bands to rank 0 through the library's communicator, so that a multi-GPU run also measures a transfer.

Other workloads (parity-tested configs of BASELINE.json, reported in DESIGN.md):
    --workload mix1           config #1: one Mix(Add) node, two 4096^2 f32x4 inputs
    --workload resize_blend   config #2: 512^2 -> 4096^2 Triangle resize + 3-node blend chain
    --workload chain32 --size 8192   config #3 at its full size (with N > 1: an independent graph per GPU, weak scaling)
    --workload chain32_rows --size 8192   config #3 split by row bands over the ranks through the library's band path
                              (kc_live_graph_evaluate_band; strong scaling, no exchange)
    --workload fanin          config #4 (any N, also N = 1): 8 independent 16-node subgraphs + a 7-node Mix(Add) tree as ONE
                              graph; the library's partitioner (kc_live_graph_partition, --policy auto | spread | bands)
                              keeps it on one GPU, places the branches on the ranks (results sent to the home rank) or gives
                              every rank its row band of the whole graph (finished bands gathered on the home rank); the
                              library's communicator moves the data; "plan" says which ran, with and without the gather
    --workload e2e            the host boundary (row f-4): RGBA8 host images in (deconstruct_image) -> the 32-node graph -> RGBA8 out
                              (to_u8) through the pipelined u8 route (kc_u8_pipe: pinned buffers, copy streams; the upload of image
                              k + 1 and the download of image k - 1 overlap the evaluation of image k); images/s and the fraction of
                              the box's measured PCIe rate ("pcie")
    --cold                    (chain32 / mix1 / resize_blend, N = 1) the timed region itself rotates over 4 instances on different
                              inputs: the run whose kernel trace reproduces roofline.frac_cold
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


def usable_cores():
    """Host cores this process may really use: the affinity mask and the cgroup CPU quota count, not just what the box has."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period))))
            break
        except (OSError, ValueError):
            continue
    return n


def embed(kc, lg, image, eid):
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, image), eid)
    return lg.add_node(kc.Node.new(kc.NodeType.Embed(eid)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=["chain32", "chain32_rows", "mix1", "resize_blend", "fanin", "e2e"],
                    help="default: chain32 at 4096^2 (the headline) on one GPU; on several, chain32_rows at 8192^2 = BASELINE "
                         "config #3, the 32-node graph split by row bands through the library's band path")
    ap.add_argument("--policy", default="auto", choices=["spread", "auto", "bands"],
                    help="fanin: placement policy of the partitioner (auto = the cheapest of one GPU / branches / row bands + gather)")
    ap.add_argument("--cold", action="store_true",
                    help="the timed region rotates over 4 instances of the workload on different inputs (nothing of a step is left in "
                         "the Infinity Cache): what roofline.frac_cold measures, as the main region -- for profiling that case")
    ap.add_argument("--size", type=int, default=None, help="default 4096; 8192 for the multi-GPU default workload")
    ap.add_argument("--nodes", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the unfused and PCIe-inclusive side measurements")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier and the reductions of the timing (nccl = RCCL; gloo to rehearse "
                         "N > 1 with several processes on one GPU); plane data moves through the library's communicator either way")
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
    elif args.workload == "e2e":
        assert world == 1, "e2e is a single-GPU workload"
        depth = 3
        pipe = kc.U8Pipe(S, S, 4, depth)
        rng = np.random.default_rng(0x5EED)
        host_imgs = [rng.integers(0, 256, size=(S, S, 4), dtype=np.uint8) for _ in range(depth)]
        for sl in range(depth):
            pipe.in_buffer(sl)[...] = host_imgs[sl]  # a real job decodes its files straight into these pinned buffers
        host_b = synth(SEED_B, S)
        img_b = kc.SlotImage.from_planes(host_b)
        white = kc.SlotImage.from_value((S, S), 1.0, True)

        def graph_on(img):
            # the 32-node graph of the headline through the per-node operator boundary (mix::process, src/node/mix.rs:51-134)
            x = img
            for i in range(1, N + 1):
                if i & 1:
                    x = kc.mix_process(x, img_b, kc.MixType.Multiply if (i >> 1) & 1 else kc.MixType.Add)
                else:
                    x = kc.mix_process(white, x, kc.MixType.Subtract)
            return x

        e2e = {"k": 0, "img": pipe.upload(0), "last": None}

        def step():
            k = e2e["k"]
            sl = k % depth
            res = graph_on(e2e["img"])
            if k >= depth:
                pipe.wait_download(sl)  # the slot's previous image must have reached the host (and been consumed) first
            pipe.download(sl, res)      # forces the chain: one fused launch + to_u8, then the copy on the pipe's stream
            e2e["img"] = pipe.upload((k + 1) % depth)  # next image: its copy runs during the kernels just enqueued
            e2e["last"] = (sl, k % depth)
            e2e["k"] = k + 1

        g = None
        node_px = float(N) * S * S
        # per image: from_u8 (4 B read, 16 B written per pixel), the fused graph (36 B), to_u8 (12 B read -- alpha is a constant -- 4 B written)
        alg_bytes = (20.0 + 36.0 + 16.0) * S * S
        kernel = "from_u8_kernel + kc_chain_<hash> + to_u8_kernel per image"
        desc = ("RGBA8 host image in -> %d-node graph at %dx%d -> RGBA8 host image out, pipelined u8 route (kc_u8_pipe, depth %d)" % (N, S, S, depth))
    else:  # fanin
        # BASELINE config #4 as ONE graph that every rank builds: 8 independent 16-node subgraphs + a fixed-order 7-node
        # Mix(Add) tree.  The library's partitioner (csrc/partition.cpp) decides -- one GPU, branches placed on the ranks with
        # the join on the home rank, or every rank its row band of the whole graph + a gather of the finished bands -- and the
        # library's communicator moves what has to move (csrc/comm.cpp).  Each rank holds only the data its plan gives it.
        from kanter_core_amd.multi_gpu import PartitionedEvaluator
        n_branches, sub_nodes = 8, 16

        def build_fanin(lg):
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
            return srcs, firsts, lasts, level[0]

        policy = {"spread": kc.PartitionPolicy.Spread, "auto": kc.PartitionPolicy.Auto, "bands": kc.PartitionPolicy.Bands}[args.policy]
        # the plan, from a probe graph whose sources are constant placeholders of the right size (no HBM behind them)
        probe = tp.new_live_graph()
        build_fanin(probe)
        for e in range(2 * n_branches):
            probe.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_value((S, S), 0.0, True)), e)
        lg = tp.new_live_graph()
        srcs, firsts, lasts, root = build_fanin(lg)
        probe_plan = probe.partition(root, world, policy)
        seeds = lambda k: (0x5EED0100 + k, 0x5EED0200 + k)  # noqa: E731
        if probe_plan.kind == kc.PlanKind.Bands:
            fy0, fy1 = probe_plan.bands[rank]
            need = probe.band_source_rows(root, fy0, fy1)
            for k in range(n_branches):
                for j, (node, seed) in enumerate(zip(srcs[k], seeds(k))):
                    a_, b_, _, fh = need[node]
                    planes = [splitmix_rows(seed, c, S, S, a_, b_) for c in range(4)]  # pointwise graph: no wrapped rows
                    lg.embed_slot_data_band(kc.SlotData(0, 0, kc.SlotImage.from_planes(planes)), 2 * k + j, a_, fh)
            mine = list(range(n_branches))
            my_rows = fy1 - fy0
        else:
            placed = {n: r for (n, r, _, _) in probe_plan.nodes}
            mine = [k for k in range(n_branches) if placed[lasts[k]] == rank]
            for k in range(n_branches):
                for j, (node, seed) in enumerate(zip(srcs[k], seeds(k))):
                    if placed[node] == rank:
                        planes = [splitmix_plane(seed, c, S, S) for c in range(4)]
                        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(planes)), 2 * k + j)
            my_rows = S
        del probe
        ev = PartitionedEvaluator(lg, root, policy=policy)
        assert ev.plan.kind == probe_plan.kind and ev.plan.transfers == probe_plan.transfers and ev.plan.bands == probe_plan.bands
        g = (lg, None, None, root)
        keep = []

        def step():
            if ev.plan.kind != kc.PlanKind.Bands:
                for k in mine:
                    lg.connect(srcs[k][0], firsts[k], 0, 0)  # re-plugging the input dirties the branch and what joins it
            keep[:] = [ev.evaluate()]

        n_tree = n_branches - 1
        if ev.plan.kind == kc.PlanKind.Bands:
            node_px = float(n_branches * sub_nodes + n_tree) * S * my_rows
        else:
            node_px = float(len(mine) * sub_nodes + (n_tree if rank == ev.plan.home else 0)) * S * S
        # an upper bound only: what the launches really move is decided at run time and is what the library counts -- the
        # roofline below uses that count (algorithmic_bytes_per_step)
        alg_bytes = len(mine) * 36.0 * S * my_rows + ((15 * 3 * 4.0 * S * my_rows) if rank == ev.plan.home else 0.0)
        kernel = "kc_chain_<hash> (config #4's programs compiled to straight-line code, csrc/specialize.cpp)"
        desc = ("8 independent 16-node subgraphs + 7-node Mix(Add) tree at %dx%d f32x4, BASELINE config #4; plan of kc_live_graph_partition "
                "(policy %s): %s" % (S, S, args.policy, ("one GPU", "branches, results sent to the home rank", "row bands + gather on the home rank")[ev.plan.kind]))

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

    first_sightings = {}  # what the steps before a compile landed took (only when this process had to compile)

    def warm(stepfn, warmup):
        # Compiled kernels outlive the process (csrc/specialize.cpp, kernel cache: the build pre-compiles the BASELINE programs,
        # every other program is written to ~/.cache/kanter_core_amd by the process that first compiles it), so normally the first
        # evaluation already runs the kernel of its program and the warm-up is EXACTLY the `warmup` steps asked for.
        # Only when this process has to compile (a fresh cache): the first sightings run through the interpreter while hiprtc
        # works on a worker thread (~2 s for the first program, GPU idle, clocks down), so the steps wait for the compile and a
        # few more untimed steps bring the clocks back -- counted and reported (extra_warmup_after_kernel_compile), along with
        # what the interpreter steps took.
        c0 = kc.specialize_stats()
        done = 0
        for _ in range(min(2, warmup)):
            stepfn()
            done += 1
        c1 = kc.specialize_stats()
        compiling = world == 1 and (c1["kernels_compiled"] > c0["kernels_compiled"] or c1["compiles_pending"] > 0)
        if world > 1:  # every rank must run the same number of steps: the ranks agree on whether anybody compiles
            flag = torch.tensor([float(c1["kernels_compiled"] > c0["kernels_compiled"] or c1["compiles_pending"] > 0)], device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            compiling = flag.item() > 0
        extra = 0
        if compiling:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            stepfn()
            stepfn()
            e1.record(stream)
            torch.cuda.synchronize()
            first_sightings["ms_per_step"] = round(e0.elapsed_time(e1) / 2, 4)
            first_sightings["note"] = "steps of this run before its kernel had been compiled (interpreter / cut chains)"
            extra += 2
            for _ in range(3):  # programs that join chains are first built at the evaluation after the one that queued them
                kc.specialize_wait()
                before = kc.specialize_stats()["kernels_compiled"]
                stepfn()
                stepfn()
                extra += 2
                kc.specialize_wait()
                if kc.specialize_stats()["kernels_compiled"] == before:
                    break
            for _ in range(40):
                stepfn()
            extra += 40
        extra_warmup[0] += extra
        for _ in range(warmup - done):
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

    def measure_pcie(mb=64, reps=16):
        """The box's PCIe rates with pinned memory: H2D alone, D2H alone, both at once (GB/s per direction)."""
        n = mb << 20
        hin, hout = torch.empty(n, dtype=torch.uint8, pin_memory=True), torch.empty(n, dtype=torch.uint8, pin_memory=True)
        din, dout = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")
        s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()

        def run(up, down):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                if up:
                    with torch.cuda.stream(s_up):
                        din.copy_(hin, non_blocking=True)
                if down:
                    with torch.cuda.stream(s_dn):
                        hout.copy_(dout, non_blocking=True)
            torch.cuda.synchronize()
            return n * reps / (time.perf_counter() - t0) / 1e9

        run(True, True)
        best = lambda up, down: max(run(up, down) for _ in range(3))  # the peak is what the link CAN do: the best of three runs
        return {"h2d_GBps": round(best(True, False), 1), "d2h_GBps": round(best(False, True), 1),
                "duplex_GBps_per_direction": round(best(True, True), 1),
                "how": "%d MB pinned copies (the size of one RGBA8 4096^2 image), %d in a row per direction, best of 3 runs" % (mb, reps)}

    main_step = step
    if args.cold:
        assert world == 1 and band is None and args.workload in ("chain32", "mix1", "resize_blend"), "--cold: single-GPU chain32 / mix1 / resize_blend"
        rot = [step] + [make_cold(i) for i in range(3)]
        rot_i = [0]

        def main_step():
            rot[rot_i[0] % len(rot)]()
            rot_i[0] += 1
        args.steps = max(len(rot), args.steps // len(rot) * len(rot))
        args.warmup = max(2 * len(rot), args.warmup // len(rot) * len(rot))
    wall, dev_s, launches = timed(main_step, args.steps, args.warmup)
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
    main_step_us = step_spread(main_step, max(20, min(args.steps, 100)))
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
    pmc = os.path.join(ROOT, "profiles", "r04_pmc_upsample_chain_kernel.json" if args.workload == "resize_blend" else "r04_pmc_chain_kernel.json")
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
        "kernel_cache": kc.kernel_cache_stats(),
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
            "timed_region": ("4 instances of the workload on different inputs in rotation (--cold): nothing a step touches is left in the Infinity Cache"
                             if args.cold else "one graph re-evaluated on the same resident inputs (the contract's step)"),
            "frac_includes_infinity_cache_hits": bool(kc.get_cache_policy()) and not args.cold,
            "parallelism": (("row bands of one graph through kc_live_graph_evaluate_band, no exchange" if band is not None else
                             "independent graph per GPU") if args.workload != "fanin"
                            else "plan of the library's partitioner (see \"plan\"), data moved by the library's communicator")
            if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": peak_gbs, "unit": "GB/s",
            "frac": round(achieved / peak_gbs, 4), "frac_cold": round(achieved / peak_gbs, 4) if args.cold else None,
            "frac_definition": ("algorithmic bytes of a step / HIP-event time of a step / (n_gpus x 8 TB/s).  frac: the timed region as the contract "
                                "defines it -- ONE graph re-evaluated on the same resident inputs; with cache_policy = 1 one long-lived input "
                                "(192 MB) can stay in the 256 MB Infinity Cache from step to step, so frac is a fabric-side figure that can exceed "
                                "what HBM alone delivers (6.3 TB/s measured copy = 0.79).  frac_cold: the same steps rotated over 4 instances on "
                                "different inputs -- everything comes from HBM -- the HBM-only fraction"),
            "traffic": traffic, "traffic_source": traffic_source,
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

    if first_sightings:
        out["first_sightings"] = dict(first_sightings)
    if args.workload == "e2e":
        for sl in range(depth):
            pipe.wait_download(sl)
        pcie = measure_pcie()
        per_image_s = wall / args.steps
        in_b, out_b = 4.0 * S * S, 4.0 * S * S
        bound_s = max(in_b, out_b) / (pcie["duplex_GBps_per_direction"] * 1e9)  # both directions at once, each at its duplex rate
        out["metric"] = "images/s, RGBA8 %dx%d in -> %d-node graph -> RGBA8 out (host to host)" % (S, S, N)
        out["value"] = round(1.0 / per_image_s, 1)
        out["unit"] = "images/s"
        out["roofline"] = {"bound": "pcie", "achieved": round(in_b / per_image_s / 1e9, 2), "peak": pcie["duplex_GBps_per_direction"],
                           "unit": "GB/s", "frac": round(bound_s / per_image_s, 4),
                           "frac_definition": "time the slower direction of one image needs at the measured full-duplex PCIe rate / measured time per image",
                           "gpu_side": {"algorithmic_bytes_per_image": alg_bytes, "hbm_time_us_at_6.1TBps": round(alg_bytes / 6.1e12 * 1e6, 1)},
                           "launches_per_step": launches_per_step, "traffic": None}
        out["pcie"] = pcie
        out["bytes_per_image"] = {"in": in_b, "out": out_b, "f32_route_would_move": 2 * 16.0 * S * S}
        if not args.no_cpu_baseline:
            # parity: what reached the host for the last image of every slot against the oracle's deconstruct -> graph -> to_u8 on
            # three 32-row crops (the graph is pointwise)
            from oracle import oracle as orc
            orc.set_threads(usable_cores())
            mism = checked = 0
            for sl in range(depth):
                got = pipe.out_buffer(sl)
                for r0 in (0, S // 2 - 16, S - 32):
                    a_pl = orc.deconstruct_u8(host_imgs[sl][r0:r0 + 32])
                    ref = orc.chain32(a_pl, [p[r0:r0 + 32] for p in host_b], N)
                    want_u8 = orc.to_u8(orc.Image(ref))
                    mism += int((got[r0:r0 + 32] != want_u8).sum())
                    checked += 32 * S * 4
            orc.set_threads(1)
            out["parity"] = {"checked_bytes": checked, "byte_mismatches": mism}
        pipe.close()
    if rank == 0 and world == 1 and band is None and args.workload in ("chain32", "mix1", "resize_blend") and not args.no_extras and not args.cold:
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
        out["roofline"]["frac_cold"] = round(counted[0] * kc_steps / dev_c / 1e9 / HBM_PEAK_GBS, 4)
        out["roofline"]["cold"] = {
            "kernel_us": round(dev_c / max(launches_c, 1) * 1e6, 2), "frac": round(counted[0] * kc_steps / dev_c / 1e9 / HBM_PEAK_GBS, 4),
            "steps": kc_steps,
            "how": "4 instances of the workload on different inputs evaluated in rotation: inputs and results of a step were last "
                   "touched more than 1.5 GB of traffic earlier, so the 256 MB Infinity Cache holds none of them",
        }
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
        moved = (8 + 4) * 4.0 * S * rows
        out["pcie_inclusive"] = {"value": round(node_px / pcie_s / 1e6, 1), "unit": "Mpix/s", "seconds": round(pcie_s, 4),
                                 "bytes_moved": moved, "GBps": round(moved / pcie_s / 1e9, 1),
                                 "note": "the f32 route, blocking: 8 pageable host planes uploaded, 4 downloaded, graph built and evaluated "
                                         "once (12 x 64 MB over PCIe, one direction at a time).  The u8 route moves a quarter of that and "
                                         "overlaps the directions: --workload e2e gives its rate and its fraction of the box's PCIe peak"}
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
        # the same loops, rows split over host cores (SURVEY 8(d) ii): every core of the box, and one GPU's share of them (1/8)
        def cpu_many(threads, pool):
            orc.set_threads(threads)
            orc.set_plane_pool(pool)
            t0 = time.perf_counter()
            orc.chain32(host_a, host_b, N)  # thread start-up, first touch
            first = time.perf_counter() - t0
            t0, n = time.perf_counter(), 0
            while n < 1 or (n < 3 and first < 4.0) or (time.perf_counter() - t0 < 3.0 and n < 12 and first < 1.0):
                orc.chain32(host_a, host_b, N)
                n += 1
            dt = time.perf_counter() - t0
            orc.set_threads(1)
            orc.set_plane_pool(False)
            return round(float(N) * S * S * n / dt / 1e6, 2), n, dt

        box_cores, all_cores = os.cpu_count() or 1, usable_cores()
        if all_cores > 16:
            # A CPU share can be enforced where neither the affinity mask nor cpu.max shows it (the GPU boxes of this pool give one
            # GPU's job 16 of 256 cores): 256 threads on 16 cores run at a twentieth of the speed.  A small probe decides.
            pa, pb = [p[:1024, :1024].copy() for p in host_a], [p[:1024, :1024].copy() for p in host_b]

            def probe(threads):
                orc.set_threads(threads)
                orc.set_plane_pool(True)
                orc.chain32(pa, pb, N)
                t0 = time.perf_counter()
                orc.chain32(pa, pb, N)
                dt = time.perf_counter() - t0
                orc.set_threads(1)
                orc.set_plane_pool(False)
                return dt

            t_all, t_share = probe(all_cores), probe(16)
            if t_share < t_all:
                out["cpu_threads_probe"] = {"threads_%d_s" % all_cores: round(t_all, 3), "threads_16_s": round(t_share, 3),
                                            "note": "1024x1024 probe: more threads than the job may run at once are slower; 16 used"}
                all_cores = 16
        if all_cores > 1:
            v, n, dt = cpu_many(all_cores, True)
            out["cpu_baseline_all_cores"] = {
                "value": v, "unit": "Mpix/s", "cores": all_cores, "kind": "port", "host_cores_in_the_box": box_cores,
                "note": "a reported baseline, not a target.  Rows split over every host core this process may use (affinity mask and "
                        "cgroup CPU quota: %d of the box's %d); planes released by a node are reused by the next one "
                        "(oracle.set_plane_pool) -- with the reference's allocate-per-node the page faults of the fresh 64 MiB mappings "
                        "serialise in the kernel and the run does not scale (cpu_baseline_per_node_alloc keeps that form)" % (all_cores, box_cores),
                "sample": "%d evaluations (%.1f s)" % (n, dt)}
            v, n, dt = cpu_many(all_cores, False)
            out["cpu_baseline_per_node_alloc"] = {
                "value": v, "unit": "Mpix/s", "cores": all_cores, "kind": "port",
                "note": "the same threads with the reference's per-node plane allocation kept", "sample": "%d evaluations (%.1f s)" % (n, dt)}
        # parity of the timed workload against the oracle, on the same inputs
        got = g[0].slot_data(g[3], 0).image.planes()
        mism = int(sum((x.view(np.uint32) != y.view(np.uint32)).sum() for x, y in zip(got, ref)))
        out["parity"] = {"checked_pixels": S * rows * 4, "bit_mismatches": mism}

    if rank == 0 and world == 1 and args.workload in ("mix1", "resize_blend", "fanin") and not args.no_cpu_baseline and g is not None:
        # parity of the timed graph against the oracle on the same inputs (the oracle as checker, after the timed region)
        from oracle import oracle as orc
        orc.set_threads(usable_cores())
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
            dist.all_gather_object(allst, st)
        else:
            allst = [st]
        out["per_rank"] = [{k: (round(v, 6) if isinstance(v, float) else v) for k, v in x.items()} for x in allst]
        out["plan"] = {"kind": ("single", "branches", "bands")[ev.plan.kind], "policy": args.policy,
                       "estimates_in_units_of_one_fused_rgba_chain": ev.plan.estimates, "bands": ev.plan.bands or None,
                       "transfers": len(ev.plan.transfers), "transport": kc.comm_transport() or None}
        if ev.plan.kind == kc.PlanKind.Bands and world > 1:
            # the same plan with the bands left where they are (a consumer that is row-parallel too): what the gather costs
            ev.plan.set_gather(False)
            wall_ng, dev_ng, _ = timed(step, max(5, args.steps // 2), 2)
            ev.plan.set_gather(True)
            t = torch.tensor([wall_ng], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out["plan"]["ms_per_step_without_gather"] = round(float(t[0].item()) / max(5, args.steps // 2) * 1e3, 4)
        out["multi_gpu_measured"] = world > 1 and args.dist_backend == "nccl"

    if band is not None and world > 1:
        # The default N > 1 workload moves no plane (a pointwise graph by rows needs no exchange).  This leg does: every rank's
        # finished band goes to rank 0's row offset through the library's communicator (kc_comm_gather_bands), so that a multi-GPU
        # run also carries a measured transfer.  Outside the timed region above; never part of `value`.
        leg = {}
        # This leg is the first thing in the run that makes the ranks wait for EACH OTHER'S GPUs (counters in shared memory that
        # streams wait on): if it ever stalls, the line measured above must still come out.  Every rank arms the same timer; when
        # it fires rank 0 prints the line with the leg marked as timed out and all ranks leave without tearing anything down.
        import threading

        def _give_up():
            if rank == 0:
                out["gather_to_rank0"] = {"error": "timed out after %d s (the leg was abandoned; everything else in this line was measured before it)" % leg_limit}
                print(json.dumps(out), flush=True)
            os._exit(0)

        leg_limit = int(os.environ.get("KC_BENCH_LEG_TIMEOUT_S", "150"))
        watchdog = threading.Timer(leg_limit, _give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            from kanter_core_amd.multi_gpu import ensure_communicator
            ensure_communicator(rank, world)
            lgb, _, _, lastb = g
            k_g = max(3, min(args.steps, 20))

            def gather_step():
                b_img = lgb.evaluate_band(lastb, band[0], band[1])
                band_keep[:] = [kc.comm_gather_bands(b_img, band[0], S, 0)]

            wall_g, dev_g, _ = timed(gather_step, k_g, 2)
            t = torch.tensor([wall_g], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            moved = 3.0 * 4 * S * (S - (multi_gpu.row_bands(S, world)[0][1] - multi_gpu.row_bands(S, world)[0][0]))
            leg = {"ms_per_step": round(float(t[0].item()) / k_g * 1e3, 4), "steps": k_g, "transport": kc.comm_transport(),
                   "bytes_into_rank0_per_step": moved, "inbound_GBps": round(moved / (float(t[0].item()) / k_g) / 1e9, 1),
                   "what": "band evaluation + gather of the R, G, B rows of the other ranks' bands into rank 0's image (alpha is a constant)"}
            if rank == 0 and not args.no_cpu_baseline:
                from oracle import oracle as orc
                got = band_keep[0].planes()
                mism = 0
                for r0 in (0, S // 2 - 16, S - 32):  # crops in the first, a middle and the last band
                    ha = [splitmix_rows(SEED_A, c, S, S, r0, r0 + 32) for c in range(4)]
                    hb = [splitmix_rows(SEED_B, c, S, S, r0, r0 + 32) for c in range(4)]
                    ref = orc.chain32(ha, hb, N)
                    mism += int(sum((got[c][r0:r0 + 32].view(np.uint32) != np.ascontiguousarray(ref[c]).view(np.uint32)).sum() for c in range(4)))
                leg["parity"] = {"checked_pixels": 3 * 32 * S * 4, "bit_mismatches": mism}
            band_keep[:] = []
        except Exception as e:  # noqa: BLE001 -- the headline line must survive a failing side leg
            leg = {"error": "%s: %s" % (type(e).__name__, e)}
        out["gather_to_rank0"] = leg
        if "error" in leg:
            # a rank that failed left the others waiting in the leg's collectives: nobody can reach the barrier below together
            watchdog.cancel()
            if rank == 0:
                print(json.dumps(out), flush=True)
            os._exit(0)
        watchdog.cancel()

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
