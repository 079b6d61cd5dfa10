"""CPU soak of the multi-rank exchange loop (multi_gpu.PartitionedEvaluator) over gloo: `world` processes evaluate the same
random, type-correct graphs with the oracle as slot store; the home rank's result must equal a single-process evaluation.
    python profiles/soak_exchange.py [world] [graphs]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp


def random_graph(rng):
    from golden_graphs import G
    OPS = ["Add", "Subtract", "Multiply", "Divide"]
    g = G()
    outs = []  # (node, slot, "R" | "G")
    for e in range(int(rng.integers(1, 4))):
        outs.append((g.add({"Embed": e}), 0, "R"))
    if rng.random() < 0.5:
        outs.append((g.add({"Value": float(rng.random())}), 0, "G"))
    for _ in range(int(rng.integers(2, 30))):
        k = rng.integers(7)
        def pick(t=None):
            c = [o for o in outs if t is None or o[2] == t]
            return c[rng.integers(len(c))] if c else None
        if k <= 3:
            l, r = pick(), pick()
            n = g.add({"Mix": OPS[rng.integers(len(OPS))]})
            g.connect(l[0], n, l[1], 0)
            if rng.random() < 0.9:
                g.connect(r[0], n, r[1], 1)
            outs.append((n, 0, l[2]))
        elif k == 4:
            o = pick("R")
            if o:
                n = g.add("SeparateRgba"); g.connect(o[0], n, o[1], 0)
                outs += [(n, s, "G") for s in range(4)]
        elif k == 5:
            if pick("G"):
                n = g.add("CombineRgba")
                for s in range(4):
                    if rng.random() < 0.8:
                        o = pick("G"); g.connect(o[0], n, o[1], s)
                outs.append((n, 0, "R"))
        else:
            o = pick("G")
            if o:
                n = g.add("HeightToNormal"); g.connect(o[0], n, o[1], 0)
                outs.append((n, 0, "R"))
    return g.dict(), outs[-1][0], outs[-1][1]


def worker(rank, world, port, n_graphs, q, device=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import kanter_core_amd as kc
    if device:
        import torch
        kc.init(0)  # every rank on cuda:0: a rehearsal of the RCCL path with planes staged through the host (gloo)
    import test_multi_gpu_gloo as mg
    from kanter_core_amd.multi_gpu import PartitionedEvaluator
    from oracle import oracle as orc
    rng = np.random.default_rng(0x50AC0004)
    bad = 0
    for gi in range(n_graphs):
        graph, root, slot = random_graph(rng)
        policy = kc.PartitionPolicy.Spread if gi & 1 else kc.PartitionPolicy.Auto
        emb = mg.embedded_images(orc, graph)
        if device:
            plan = mg.host_live_graph(graph).partition(root, world, policy)
            mine = {n for (n, r, _, k) in plan.nodes if r == rank and k == kc.NodeKind.Source}
            tp = kc.TextureProcessor.new()
            lg = tp.new_live_graph()
            lg.set_node_graph(kc.NodeGraph.from_json(json.dumps(graph)))
            for nd in graph["nodes"]:
                t = nd["node_type"]
                if isinstance(t, dict) and "Embed" in t and nd["node_id"] in mine:
                    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(emb[t["Embed"]].planes)), t["Embed"])
            ev = PartitionedEvaluator(lg, root, policy=policy, device=torch.device("cuda", 0))
            img = ev.evaluate()
            got = img.planes() if img is not None else None
        else:
            lg = mg.host_live_graph(graph)
            ev = PartitionedEvaluator(lg, root, policy=policy, backend=lambda plan, r: mg.OracleBackend(orc, graph, emb, plan, r))
            for rep in range(2):
                ev.backend.ref.results.clear()
                img = ev.evaluate()
            got = img.planes if img is not None else None
        if rank == ev.plan.home:
            want = orc.RefGraph(graph, embedded=emb).slot_data(root, 0).image.planes
            nan_eq = lambda a, b: a.shape == b.shape and bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
            ok = got is not None and len(got) == len(want) and all(nan_eq(np.asarray(a), np.asarray(b)) for a, b in zip(got, want))
            if not ok:
                bad += 1
                print("MISMATCH graph %d world %d" % (gi, world), json.dumps(graph)[:800], flush=True)
        elif img is not None:
            bad += 1
            print("rank %d got a result for graph %d" % (rank, gi), flush=True)
    q.put((rank, bad))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    device = len(sys.argv) > 3 and sys.argv[3] == "device"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    procs = [ctx.Process(target=worker, args=(r, world, port, n, q, device)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=3000) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    bad = sum(b for _, b in res)
    print("exchange soak: world %d, %d graphs, %d failures, %.0f s, exit codes %s" % (world, n, bad, time.time() - t0, [p.exitcode for p in procs]))
    sys.exit(1 if bad or any(p.exitcode for p in procs) else 0)
