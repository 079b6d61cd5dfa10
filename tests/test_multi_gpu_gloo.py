"""The N > 1 path on CPU: world_size-2 and -3 `gloo` processes run the library's partitioner
(kc_live_graph_partition, host-only: no GPU needed) on real NodeGraph JSON -- a graph with a diamond, a fan-in
and a slot that has consumers on two ranks (a broadcast) -- and the exchange loop of multi_gpu.PartitionedEvaluator
with REAL pixel work on every rank: the slot store is the CPU oracle's literal process_node (tests may use the
oracle as a checker), planes cross rank boundaries over gloo, and the result on the home rank must equal a
single-process evaluation bit for bit.  A rank that touched a node the plan did not place on it fails loudly: each
rank only holds the source images the plan gives it."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import kanter_core_amd as kc
from golden_graphs import G
from kanter_core_amd.multi_gpu import PartitionedEvaluator, row_bands
from util import splitmix_plane

H, W = 20, 24


def test_row_bands_cover_the_plane():
    assert row_bands(8192, 2) == [(0, 4096), (4096, 8192)]
    assert row_bands(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert row_bands(100, 3, align=16) == [(0, 48), (48, 80), (80, 100)]
    for h in (1, 7, 4096, 8191):
        for w in (1, 2, 3, 8):
            bands = row_bands(h, w)
            assert bands[0][0] == 0 and bands[-1][1] == h
            assert all(bands[i][1] == bands[i + 1][0] for i in range(w - 1))


# ------------------------------------------------------------------------------------------ graphs
def diamond_fanin_broadcast_graph():
    """E0, E1, E2 embedded RGBA images.
       branch A:  a1 = E0 + E1;  a2 = 1 - a1
       prefix P:  p1 = E2 * E2                          (fan-out: consumed by b1 AND c1 -> a broadcast when they
       branch B:  b1 = p1 + E0                           land on different ranks; E0 is also read by branch A)
       branch C:  c1 = p1 - E1;  c2 = HeightToNormal(Separate(c1).R)
       joins:     j1 = a2 + b1;  j2 = j1 * c2  (diamond over p1);  out = OutputRgba(j2)"""
    g = G()
    e = [g.add({"Embed": i}) for i in range(3)]
    one = g.add({"Value": 1.0})
    white = g.add("CombineRgba")
    for s in range(3):
        g.connect(one, white, 0, s)
    a1 = g.add({"Mix": "Add"})
    g.connect(e[0], a1, 0, 0)
    g.connect(e[1], a1, 0, 1)
    a2 = g.add({"Mix": "Subtract"})
    g.connect(white, a2, 0, 0)
    g.connect(a1, a2, 0, 1)
    p1 = g.add({"Mix": "Multiply"})
    g.connect(e[2], p1, 0, 0)
    g.connect(e[2], p1, 0, 1)
    b1 = g.add({"Mix": "Add"})
    g.connect(p1, b1, 0, 0)
    g.connect(e[0], b1, 0, 1)
    c1 = g.add({"Mix": "Subtract"})
    g.connect(p1, c1, 0, 0)
    g.connect(e[1], c1, 0, 1)
    sep = g.add("SeparateRgba")
    g.connect(c1, sep, 0, 0)
    c2 = g.add("HeightToNormal")
    g.connect(sep, c2, 0, 0)
    j1 = g.add({"Mix": "Add"})
    g.connect(a2, j1, 0, 0)
    g.connect(b1, j1, 0, 1)
    j2 = g.add({"Mix": "Multiply"})
    g.connect(j1, j2, 0, 0)
    g.connect(c2, j2, 0, 1)
    out = g.add({"OutputRgba": "out"})
    g.connect(j2, out, 0, 0)
    names = dict(e0=e[0], e1=e[1], e2=e[2], one=one, white=white, a1=a1, a2=a2, p1=p1, b1=b1, c1=c1, sep=sep, c2=c2,
                 j1=j1, j2=j2, out=out)
    return g.dict(), out, names


def fanin_graph(n_branches=8, n_nodes=4):
    """BASELINE config #4's shape at toy size: n independent chains (own sources 2k, 2k+1) + a Mix(Add) tree."""
    g = G()
    lasts = []
    for k in range(n_branches):
        a, b = g.add({"Embed": 2 * k}), g.add({"Embed": 2 * k + 1})
        one = g.add({"Value": 1.0})
        white = g.add("CombineRgba")
        for s in range(3):
            g.connect(one, white, 0, s)
        prev = a
        for i in range(1, n_nodes + 1):
            if i & 1:
                n = g.add({"Mix": "Multiply" if (i >> 1) & 1 else "Add"})
                g.connect(prev, n, 0, 0)
                g.connect(b, n, 0, 1)
            else:
                n = g.add({"Mix": "Subtract"})
                g.connect(white, n, 0, 0)
                g.connect(prev, n, 0, 1)
            prev = n
        lasts.append(prev)
    while len(lasts) > 1:
        nxt = []
        for i in range(0, len(lasts) - 1, 2):
            n = g.add({"Mix": "Add"})
            g.connect(lasts[i], n, 0, 0)
            g.connect(lasts[i + 1], n, 0, 1)
            nxt.append(n)
        if len(lasts) & 1:
            nxt.append(lasts[-1])
        lasts = nxt
    return g.dict(), lasts[0]


def embedded_images(orc, graph):
    ids = sorted(n["node_type"]["Embed"] for n in graph["nodes"] if isinstance(n["node_type"], dict) and "Embed" in n["node_type"])
    return {i: orc.Image([splitmix_plane(0x5EED0100 + i, c, H, W) for c in range(4)]) for i in ids}


class HostOnlyTexPro:
    def __init__(self):
        import ctypes as C
        from kanter_core_amd import _lib
        self._h = C.c_void_p()
        assert _lib.load().kc_tex_pro_new(10_000_000, C.byref(self._h)) == 0

    def new_live_graph(self):
        import ctypes as C
        from kanter_core_amd import _lib
        h = C.c_void_p()
        assert _lib.load().kc_tex_pro_new_live_graph(self._h, C.byref(h)) == 0
        return kc.LiveGraph(h.value, self)


def host_live_graph(graph):
    lg = HostOnlyTexPro().new_live_graph()
    lg.set_node_graph(kc.NodeGraph.from_json(json.dumps(graph)))
    return lg


# ------------------------------------------------------------------------------------------ plan properties
def check_plan(graph, root, plan, world):
    nodes = {n: (r, c, k) for (n, r, c, k) in plan.nodes}
    kinds = {n["node_id"]: n["node_type"] for n in graph["nodes"]}
    assert root in nodes and plan.world == world and plan.home == 0
    # topological: every parent of a listed node that is itself listed comes earlier
    order = [n for (n, _, _, _) in plan.nodes]
    for e in graph["edges"]:
        if e["input_id"] in nodes:
            assert e["output_id"] in nodes and order.index(e["output_id"]) < order.index(e["input_id"])
    for n, (r, c, k) in nodes.items():
        t = kinds[n]
        is_src = isinstance(t, dict) and ("Embed" in t or "Image" in t)
        assert (k == kc.NodeKind.Source) == is_src
        assert (r == -1) == (k == kc.NodeKind.Replicated) and -1 <= r < world
        if isinstance(t, dict) and "Value" in t:
            assert k == kc.NodeKind.Replicated
    # transfers == exactly the edges from a placed producer to a compute consumer on another rank
    want = set()
    for e in graph["edges"]:
        if e["input_id"] in nodes and nodes[e["input_id"]][2] == kc.NodeKind.Compute:
            pr, cr = nodes[e["output_id"]][0], nodes[e["input_id"]][0]
            if pr != -1 and pr != cr:
                want.add((e["output_id"], e["output_slot"], pr, cr))
    got = [(n, s, a, b) for (n, s, a, b, _) in plan.transfers]
    assert len(got) == len(set(got)) and set(got) == want
    # execution order: a transfer's producer never depends on a later transfer's data arriving at its rank
    levels = [lv for (_, _, _, _, lv) in plan.transfers]
    assert levels == sorted(levels) and plan.levels == (max(levels) + 1 if levels else 1)
    # ... and strictly: what a transfer's producer needs from other ranks has arrived in an EARLIER level (csrc/comm.cpp works
    # through a level's sends before its receives).  Ancestors of the producer that sit on its own rank, transitively:
    parents = {}
    for e in graph["edges"]:
        parents.setdefault(e["input_id"], []).append((e["output_id"], e["output_slot"]))
    for (n, s, src, dst, lv) in plan.transfers:
        stack, seen = [n], set()
        while stack:
            x = stack.pop()
            if x in seen:
                continue
            seen.add(x)
            for (p, ps) in parents.get(x, []):
                pr = nodes[p][0]
                if pr == -1 or pr == src:
                    stack.append(p)  # computed (or held) on the producer's rank: look further up
                else:
                    inbound = [l2 for (n2, s2, a2, d2, l2) in plan.transfers if (n2, s2, a2, d2) == (p, ps, pr, src)]
                    assert inbound and inbound[0] < lv, "transfer of node %d at level %d needs node %d, which arrives at level %s" % (n, lv, p, inbound)
    return nodes


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_partition_plan_of_diamond_fanin_broadcast_graph(world):
    graph, root, nm = diamond_fanin_broadcast_graph()
    lg = host_live_graph(graph)
    plan = lg.partition(root, world, kc.PartitionPolicy.Spread)
    nodes = check_plan(graph, root, plan, world)
    rank = lambda name: nodes[nm[name]][0]  # noqa: E731
    # chains stay together, the join region sits on the home rank, constants are replicated
    assert rank("a1") == rank("a2") and rank("c1") == rank("sep") == rank("c2")
    assert rank("j1") == rank("j2") == rank("out") == plan.home
    assert rank("one") == rank("white") == -1
    if world == 1:
        assert plan.transfers == []
    else:
        used = {rank(x) for x in ("a1", "p1", "b1", "c1")}
        assert len(used) == min(world, 3) or len(used) == min(world, 4)  # the four branch components fill the ranks
        # p1 feeds b1 and c1: whenever those sit on ranks other than p1's, p1's slot is listed once per destination
        dsts = sorted(d for (n, s, a, d, _) in plan.transfers if n == nm["p1"])
        assert dsts == sorted({rank("b1"), rank("c1")} - {rank("p1")})
    # the plan is a pure function of the graph: a second live graph gives the same answer
    again = host_live_graph(graph).partition(root, world, kc.PartitionPolicy.Spread)
    assert again.nodes == plan.nodes and again.transfers == plan.transfers


@pytest.mark.parametrize("world", [2, 4, 8])
def test_partition_plan_of_fanin_graph_is_a_gather(world):
    graph, root = fanin_graph(8, 4)
    lg = host_live_graph(graph)
    plan = lg.partition(root, world, kc.PartitionPolicy.Spread)
    nodes = check_plan(graph, root, plan, world)
    # 8 branches over `world` ranks, evenly; every transfer is a branch result going straight to the home rank
    per_rank = {}
    for (n, r, c, k) in plan.nodes:
        if k == kc.NodeKind.Compute:
            per_rank.setdefault(r, set()).add(c)
    branch_comps = {r: len(cs) for r, cs in per_rank.items()}
    assert sorted(branch_comps) == list(range(world))
    assert all(d == plan.home and lv == 0 for (_, _, _, d, lv) in plan.transfers)
    assert len(plan.transfers) == 8 - 8 // world
    # charging the transfers (Auto) keeps a graph this small on one GPU
    auto = lg.partition(root, world, kc.PartitionPolicy.Auto)
    assert auto.transfers == [] and {r for (_, r, _, k) in auto.nodes if k != kc.NodeKind.Replicated} == {0}
    assert nodes[root][0] == plan.home


def test_partition_rejects_cycles_and_bad_arguments():
    g = G()
    a, b = g.add({"Mix": "Add"}), g.add({"Mix": "Add"})
    g.connect(a, b, 0, 0)
    g.connect(b, a, 0, 0)
    lg = host_live_graph(g.dict())
    with pytest.raises(kc.TexProError):
        lg.partition(b, 2)
    graph, root = fanin_graph(2, 2)
    lg = host_live_graph(graph)
    for bad in (0, -1, 5000):
        with pytest.raises(kc.TexProError):
            lg.partition(root, bad)
    with pytest.raises(kc.TexProError):
        lg.partition(123456, 2)


# ------------------------------------------------------------------------------------------ exchange, for real
class OracleBackend:
    """Slot store for PartitionedEvaluator on the CPU: oracle.RefGraph evaluates the local nodes, torch CPU tensors
    carry the planes.  Only the embedded images of sources the plan places on this rank exist here."""

    def __init__(self, orc, graph, embedded, plan, rank):
        mine = {n for (n, r, _, k) in plan.nodes if r == rank and k == kc.NodeKind.Source}
        local = {}
        for n in graph["nodes"]:
            t = n["node_type"]
            if isinstance(t, dict) and "Embed" in t and n["node_id"] in mine:
                local[t["Embed"]] = embedded[t["Embed"]]
        self.orc, self.ref = orc, orc.RefGraph(graph, embedded=local)
        self.allowed = {n for (n, r, _, _) in plan.nodes if r in (rank, -1)}
        self.evaluated = []

    def evaluate(self, node_id):
        assert node_id in self.allowed, "rank evaluates a node that is not placed on it: %d" % node_id
        self.evaluated.append(node_id)
        self.ref.node_slot_datas(node_id)

    def export_slot(self, node_id, slot_id):
        img = self.ref.slot_data(node_id, slot_id).image
        h, w = img.planes[0].shape
        tensors = [torch.from_numpy(np.ascontiguousarray(p)) for p in img.planes]
        return {"w": w, "h": h, "planes": [("m", i) for i in range(len(tensors))]}, tensors, img

    def alloc_slot(self, header):
        n = len(header["planes"])
        tensors = [torch.empty(header["h"], header["w"], dtype=torch.float32) for _ in range(n)]
        return tensors, tensors

    def import_slot(self, node_id, slot_id, header, tensors):
        img = self.orc.Image([t.numpy() for t in tensors])
        self.ref.results.setdefault(node_id, [])
        self.ref.results[node_id] = [s for s in self.ref.results[node_id] if s.slot_id != slot_id] + [self.orc.SlotData(node_id, slot_id, img)]

    def result(self, node_id, slot_id=0):
        return self.ref.slot_data(node_id, slot_id).image


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, which, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        graph, root = (diamond_fanin_broadcast_graph()[:2] if which == "diamond" else fanin_graph(8, 4))
        lg = host_live_graph(graph)
        ev = PartitionedEvaluator(lg, root, policy=kc.PartitionPolicy.Spread,
                                  backend=lambda plan, r: OracleBackend(orc, graph, embedded_images(orc, graph), plan, r))
        for rep in range(2):  # twice: the second round must not depend on state left by the first
            ev.backend.ref.results.clear()
            img = ev.evaluate()
            assert (img is not None) == (rank == ev.plan.home)
        q.put((rank, [p.tobytes() for p in img.planes] if img is not None else None, ev.stats, ev.plan.transfers,
               sorted(set(ev.backend.evaluated))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,which", [(2, "diamond"), (3, "diamond"), (2, "fanin"), (3, "fanin")])
def test_partitioned_evaluation_over_gloo_equals_single_process(world, which):
    from oracle import oracle as orc
    graph, root = (diamond_fanin_broadcast_graph()[:2] if which == "diamond" else fanin_graph(8, 4))
    want = orc.RefGraph(graph, embedded=embedded_images(orc, graph)).slot_data(root, 0).image.planes
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, which, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    transfers = outs[0][3]
    assert all(o[3] == transfers for o in outs), "ranks disagree on the plan"
    got = outs[0][1]
    assert got is not None and all(o[1] is None for o in outs[1:])
    assert [g == w.tobytes() for g, w in zip(got, want)] == [True] * len(want)
    # every listed transfer happened, nothing else moved
    slots = {}
    for (n, s, src, dst, _) in transfers:
        slots.setdefault((n, s, src), []).append(dst)
    planes = 4  # every slot here is RGBA or gray; count by what the producers report
    sent = sum(o[2]["planes_sent"] for o in outs)
    recv = sum(o[2]["planes_received"] for o in outs)
    assert sent == recv and sent >= len(transfers) and sent <= planes * len(transfers)
    # and each rank evaluated only transfer producers placed on it (+ the root on the home rank)
    for rank, _, _, _, evaluated in outs:
        for n in evaluated:
            assert n == root or any(t[0] == n and t[2] == rank for t in transfers)


# ------------------------------------------------------------------------------------------ one GPU, branches or row bands
def test_auto_prices_one_gpu_branches_and_row_bands_for_config4():
    """BASELINE config #4 (eight 16-node branches + add tree): every node is pointwise (src/node/mix.rs:136-192), so a band plan
    moves only 1/world of the result per rank.  With the default rates (153 GB/s per link, 6.1 TB/s of HBM) that beats one GPU
    -- which runs the whole graph as ONE launch, 16 sources read once -- from four ranks on; branches never do (each branch
    result is a whole image over one link)."""
    from rank_scenarios import case
    graph, root, sizes = case("config4", 4096, 4096)
    lg = host_live_graph(graph)
    with pytest.raises(kc.TexProError):  # sizes unknown: no band plan can be cut
        lg.partition(root, 8, kc.PartitionPolicy.Bands)
    assert lg.partition(root, 8, kc.PartitionPolicy.Auto).kind == kc.PlanKind.Single  # ... and AUTO falls back
    for eid, (h, w) in sizes.items():  # constant placeholders carry the sizes: all a plan needs
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_value((w, h), 0.0, True)), eid)
    kinds = {world: lg.partition(root, world, kc.PartitionPolicy.Auto) for world in (1, 2, 3, 4, 8)}
    assert [kinds[w].kind for w in (1, 2, 3, 4, 8)] == [kc.PlanKind.Single, kc.PlanKind.Single, kc.PlanKind.Single, kc.PlanKind.Bands, kc.PlanKind.Bands]
    assert abs(kinds[8].estimates["single"] - (3 * 16 + 3) * 4 / 40) < 1e-9  # one launch: every source once + the result
    p8 = kinds[8]
    assert p8.transfers == [] and p8.full_size == (4096, 4096) and p8.bands == [(512 * r, 512 * (r + 1)) for r in range(8)]
    assert all(r == -1 for (_, r, _, _) in p8.nodes)
    assert p8.estimates["bands"] < p8.estimates["single"] < lg.partition(root, 8, kc.PartitionPolicy.Spread).estimates["branches"]
    # uneven heights: the first `height % world` ranks take one more row
    graph, root, sizes = case("config4", 37, 16)
    lg = host_live_graph(graph)
    for eid, (h, w) in sizes.items():
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_value((w, h), 0.0, True)), eid)
    assert lg.partition(root, 3, kc.PartitionPolicy.Bands).bands == [(0, 13), (13, 25), (25, 37)]
    with pytest.raises(kc.TexProError):  # fewer rows than ranks
        lg.partition(root, 16, kc.PartitionPolicy.Bands) if False else lg.partition(root, 64, kc.PartitionPolicy.Bands)
    # a slower link makes the bands lose to one GPU again, a faster one makes them win earlier
    try:
        kc.set_option("link_gbps", 20)
        assert lg.partition(root, 8, kc.PartitionPolicy.Auto).kind == kc.PlanKind.Single
        kc.set_option("link_gbps", 600)
        assert lg.partition(root, 3, kc.PartitionPolicy.Auto).kind == kc.PlanKind.Bands
    finally:
        kc.set_option("link_gbps", 153)


def test_linear_chain_is_never_gathered_by_auto():
    """BASELINE config #3's shape (one fused chain): the result's transfer alone costs more than the chain, so AUTO stays on
    one GPU; a caller whose consumer is row-parallel too asks for bands and switches the gather off."""
    g = G()
    a, b = g.add({"Embed": 0}), g.add({"Embed": 1})
    prev = a
    for i in range(8):
        n = g.add({"Mix": "Add" if i & 1 else "Multiply"})
        g.connect(prev, n, 0, 0)
        g.connect(b, n, 0, 1)
        prev = n
    lg = host_live_graph(g.dict())
    for eid in (0, 1):
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_value((8192, 8192), 0.0, True)), eid)
    assert lg.partition(prev, 8, kc.PartitionPolicy.Auto).kind == kc.PlanKind.Single
    plan = lg.partition(prev, 2, kc.PartitionPolicy.Bands)
    plan.set_gather(False)
    assert plan.kind == kc.PlanKind.Bands and plan.bands == [(0, 4096), (4096, 8192)] and not plan.gather
