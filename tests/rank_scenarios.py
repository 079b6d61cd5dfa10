"""What the ranks of tests/rank_harness.py run (imported inside the child processes; nothing here touches the oracle -- the
parent compares what comes back with the oracle's result)."""
import json

import numpy as np

from golden_graphs import G
from util import splitmix_plane


# ------------------------------------------------------------------------------------------ graphs + inputs
def config4_graph(n_branches=8, n_nodes=16, small_source_in=None, h2n_in=None):
    """BASELINE config #4: n independent chains (sources 2k, 2k+1) + a Mix(Add) tree.  small_source_in = k: branch k's second
    source is half size (an implicit Triangle up-sampling in that branch); h2n_in = k: branch k ends in
    HeightToNormal(Separate.R) (a one-row halo)."""
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
        if h2n_in == k:
            sep = g.add("SeparateRgba")
            g.connect(prev, sep, 0, 0)
            prev = g.add("HeightToNormal")
            g.connect(sep, prev, 0, 0)
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


def case(name, h, w):
    """-> (graph dict, root, {embed id: (height, width)})"""
    if name == "diamond":
        from test_multi_gpu_gloo import diamond_fanin_broadcast_graph
        graph, root, _ = diamond_fanin_broadcast_graph()
    elif name == "fanin":
        from test_multi_gpu_gloo import fanin_graph
        graph, root = fanin_graph(8, 4)
    elif name == "config4":
        graph, root = config4_graph()
    elif name == "config4_resize_h2n":
        graph, root = config4_graph(n_branches=5, n_nodes=6, small_source_in=1, h2n_in=3)
    else:
        raise KeyError(name)
    ids = sorted(n["node_type"]["Embed"] for n in graph["nodes"] if isinstance(n["node_type"], dict) and "Embed" in n["node_type"])
    sizes = {i: (h, w) for i in ids}
    if name == "config4_resize_h2n":
        sizes[3] = (h // 2, w // 2)
    return graph, root, sizes


def source_planes(eid, size):
    return [splitmix_plane(0x5EED0100 + eid, c, size[0], size[1]) for c in range(4)]


# ------------------------------------------------------------------------------------------ scenarios
def evaluate_plan(kc, rank, world, name, h, w, policy, gather=True, reps=2, specialize=None):
    """Makes the plan on a probe graph (every source a constant placeholder of its size, which is all a plan needs), builds the
    graph with REAL data only where the plan wants it -- whole sources placed on this rank (branches), the rows
    kc_live_graph_band_source_rows names (bands) -- and evaluates `reps` times."""
    if specialize is not None:
        kc.set_specialize(specialize)
    graph, root, sizes = case(name, h, w)
    embed_node = {n["node_type"]["Embed"]: n["node_id"] for n in graph["nodes"] if isinstance(n["node_type"], dict) and "Embed" in n["node_type"]}
    # the plan, from a probe graph whose sources are constant placeholders of the right size (no HBM, no data)
    tp = kc.TextureProcessor.new()
    probe = tp.new_live_graph()
    probe.set_node_graph(kc.NodeGraph.from_json(json.dumps(graph)))
    for eid, (sh, sw) in sizes.items():
        probe.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_value((sw, sh), float("nan"), True)), eid)
    probe_plan = probe.partition(root, world, policy)
    # the graph that is evaluated holds only what the plan gives this rank
    lg = tp.new_live_graph()
    lg.set_node_graph(kc.NodeGraph.from_json(json.dumps(graph)))
    if probe_plan.kind == kc.PlanKind.Bands:
        y0, y1 = probe_plan.bands[rank]
        need = probe.band_source_rows(root, y0, y1)
        for eid, node in embed_node.items():
            if node not in need:
                continue
            a, b, _, full_h = need[node]
            planes = source_planes(eid, sizes[eid])
            rows = [r % full_h for r in range(a, b)]  # a < 0: the wrapped rows come first
            lg.embed_slot_data_band(kc.SlotData(0, 0, kc.SlotImage.from_planes([p[rows] for p in planes])), eid, a, full_h)
    else:
        placed = {n: r for (n, r, _, _) in probe_plan.nodes}
        for eid, node in embed_node.items():
            if placed.get(node) == rank:
                lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(source_planes(eid, sizes[eid]))), eid)
    del probe
    plan = lg.partition(root, world, policy)
    assert plan.kind == probe_plan.kind and plan.transfers == probe_plan.transfers and plan.bands == probe_plan.bands
    if plan.kind == kc.PlanKind.Bands:
        plan.set_gather(gather)
    outs = []
    for rep in range(reps):
        img = lg.evaluate_partitioned(plan, root)
        outs.append(None if img is None else [p.tobytes() for p in img.planes()])
        if plan.kind != kc.PlanKind.Bands:  # re-dirty what this rank owns, as an editor changing the inputs would
            for (n, r, _, k) in plan.nodes:
                if r == rank and k == kc.NodeKind.Source:
                    for e in lg.edges():
                        if e.output_id == n:
                            lg.connect(e.output_id, e.input_id, e.output_slot, e.input_slot)
    return {"kind": plan.kind, "bands": plan.bands, "home": plan.home, "transfers": plan.transfers, "estimates": plan.estimates,
            "outs": outs, "stats": kc.comm_stats(), "transport": kc.comm_transport(),
            "mappings": kc.stats_counter("comm_ipc_mappings_opened")}


def gather_direct(kc, rank, world, h, w, cuts, gray, home):
    """kc_comm_gather_bands on bands this rank makes itself: rows cuts[rank] .. cuts[rank + 1] of a known image; twice, the second
    time after a pool trim on every rank (mappings of freed blocks must go)."""
    full = [splitmix_plane(0x5EED0777, c, h, w) for c in range(1 if gray else 4)]
    y0, y1 = cuts[rank], cuts[rank + 1]
    outs = []
    for rep in range(2):
        planes = [p[y0:y1] + np.float32(rep) for p in full]
        band = kc.SlotImage.from_planes(planes)
        if not gray:
            band = kc.mix_process(band, kc.SlotImage.from_value((w, y1 - y0), 0.0, True), kc.MixType.Add)  # alpha becomes the constant 1
        img = kc.comm_gather_bands(band, y0, h, home)
        assert (img is not None) == (rank == home)
        outs.append(None if img is None else [p.tobytes() for p in img.planes()])
        del img, band
        kc.sync()
        kc.pool_trim()
    return {"outs": outs, "mappings": kc.stats_counter("comm_ipc_mappings_opened")}


def mismatched_lists(kc, rank, world):
    """Rank 1 passes a transfer list naming a rank outside the communicator: it fails at once and every other rank's wait
    fails with it instead of hanging."""
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    a = [splitmix_plane(0x5EED0001, c, 16, 16) for c in range(4)]
    src = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 0)
    out = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("o")))
    lg.connect(src, out, 0, 0)
    if rank == 1:
        lg.exchange([(src, 0, 1, 99)])
    else:
        lg.exchange([(src, 0, 1, 0)])
    return "no error"


def silent_peer(kc, rank, world):
    """Rank 1 never sends what rank 0 waits for: rank 0 times out (KC_COMM_TIMEOUT_S) with an error."""
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    a = [splitmix_plane(0x5EED0001, c, 16, 16) for c in range(4)]
    src = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 0)
    if rank == 0:
        lg.exchange([(src, 0, 1, 0)])
    return "no error"


def fuzz_plans(kc, rank, world, seeds):
    """Seeded random graphs (tests/test_gpu_fuzz_graphs.py: every node type, resize policies and filters, aliasing, constants,
    sources of different sizes and types) through the partitioned path: a branch plan, and a band plan where the band walk takes
    the graph.  Every rank embeds every source here (what a rank may NOT hold is tested by evaluate_plan); what is under test is
    the plan and the exchange on arbitrary shapes: multi-level transfers, slots with consumers on several ranks, Separate's
    four slots, gray / rgba / constant planes, resizes of slots that arrived from another rank."""
    from oracle import oracle as orc  # only to let the shared builder construct its (unused) reference graph
    from test_gpu_fuzz_graphs import _build
    out = {}
    for seed in seeds:
        res = {}
        for name, policy in (("spread", kc.PartitionPolicy.Spread), ("bands", kc.PartitionPolicy.Bands)):
            lg, _, requested = _build(kc, orc, seed)
            root = int(requested[0])
            try:
                plan = lg.partition(root, world, policy)
            except kc.TexProError as e:  # host-only and deterministic: the same on every rank
                res[name] = "no plan: %s" % e.kind
                continue
            img = lg.evaluate_partitioned(plan, root)
            res[name] = {"kind": plan.kind, "transfers": len(plan.transfers), "levels": plan.levels,
                         "planes": None if img is None else [p.tobytes() for p in img.planes()],
                         "shape": None if img is None else img.planes()[0].shape}
        out[seed] = res
    return out
