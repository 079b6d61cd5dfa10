"""Partitioned (multi-rank) evaluation on the device through multi_gpu.PartitionedEvaluator as a torch.distributed program
uses it: the process group (gloo here) only carries the communicator's identifier; the exchange is the library's
(csrc/comm.cpp, IPC wire).
  * world = 1: the partitioned path is the plain evaluation; result == oracle.
  * world = 2 and 3 on ONE GPU: the processes all bound to cuda:0; the home rank's result == oracle, and the second round
    (imports replace the previous round's slots) too.
tests/test_gpu_comm_ranks.py drives the same path without torch.distributed; tests/test_multi_gpu_gloo.py the plans on CPU."""
import json
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_multi_gpu_gloo import H, W, _free_port, diamond_fanin_broadcast_graph, fanin_graph
from util import assert_planes, splitmix_plane

pytestmark = pytest.mark.gpu


def _images(graph):
    ids = sorted(n["node_type"]["Embed"] for n in graph["nodes"] if isinstance(n["node_type"], dict) and "Embed" in n["node_type"])
    return {i: [splitmix_plane(0x5EED0100 + i, c, H, W) for c in range(4)] for i in ids}


def _oracle(graph, root):
    from oracle import oracle as orc
    emb = {i: orc.Image(p) for i, p in _images(graph).items()}
    return orc.RefGraph(graph, embedded=emb).slot_data(root, 0).image.planes


def _device_graph(kc, graph, only_sources=None):
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    lg.set_node_graph(kc.NodeGraph.from_json(json.dumps(graph)))
    for n in graph["nodes"]:
        t = n["node_type"]
        if isinstance(t, dict) and "Embed" in t and (only_sources is None or n["node_id"] in only_sources):
            lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(_images(graph)[t["Embed"]])), t["Embed"])
    return tp, lg


@pytest.mark.parametrize("which", ["diamond", "fanin"])
def test_partitioned_world1_equals_oracle(which):
    import kanter_core_amd as kc
    from kanter_core_amd.multi_gpu import PartitionedEvaluator
    kc.init(0)
    graph, root = (diamond_fanin_broadcast_graph()[:2] if which == "diamond" else fanin_graph(8, 4))
    tp, lg = _device_graph(kc, graph)
    ev = PartitionedEvaluator(lg, root, device=torch.device("cuda", 0))
    assert ev.plan.transfers == [] and ev.world == 1
    assert_planes(ev.evaluate().planes(), _oracle(graph, root), what=which)


def _worker(rank, world, port, which, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kanter_core_amd as kc
        from kanter_core_amd.multi_gpu import PartitionedEvaluator
        kc.init(0)
        graph, root = (diamond_fanin_broadcast_graph()[:2] if which == "diamond" else fanin_graph(8, 4))
        # the plan first (host only), then only the sources placed on this rank get their images
        probe = kc.TextureProcessor.new().new_live_graph()
        probe.set_node_graph(kc.NodeGraph.from_json(json.dumps(graph)))
        plan = probe.partition(root, world, kc.PartitionPolicy.Spread)
        mine = {n for (n, r, _, k) in plan.nodes if r == rank and k == kc.NodeKind.Source}
        tp, lg = _device_graph(kc, graph, only_sources=mine)
        ev = PartitionedEvaluator(lg, root, policy=kc.PartitionPolicy.Spread, device=torch.device("cuda", 0))
        assert ev.plan.transfers == plan.transfers
        out = []
        for rep in range(2):
            img = ev.evaluate()
            assert (img is not None) == (rank == ev.plan.home)
            out.append([p.tobytes() for p in img.planes()] if img is not None else None)
            # re-dirty what this rank owns, as an editor changing the inputs would
            for (n, r, _, k) in ev.plan.nodes:
                if r == rank and k == kc.NodeKind.Source:
                    for e in lg.edges():
                        if e.output_id == n:
                            lg.connect(e.output_id, e.input_id, e.output_slot, e.input_slot)
        assert ev.stats["native"] and ev.stats["transport"] == "ipc"
        kc.sync()
        kc.comm_destroy()
        q.put((rank, out, ev.stats, len(ev.plan.transfers)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,which", [(2, "diamond"), (2, "fanin"), (3, "diamond"), (3, "fanin")])
def test_partitioned_on_one_gpu_equals_oracle(world, which):
    """`world` processes share cuda:0 (the pool allows 6 of ours on the card); 3 ranks give uneven branch counts and, for
    the diamond graph, a producer whose slot goes to two other ranks."""
    graph, root = (diamond_fanin_broadcast_graph()[:2] if which == "diamond" else fanin_graph(8, 4))
    want = [p.tobytes() for p in _oracle(graph, root)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, which, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert outs[0][1][0] == want and outs[0][1][1] == want and all(o[1] == [None, None] for o in outs[1:])
    assert outs[0][3] > 0 and outs[0][2]["planes_received"] > 0
    assert sum(o[2]["planes_sent"] for o in outs) == sum(o[2]["planes_received"] for o in outs)
