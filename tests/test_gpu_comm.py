"""The exchange behind the C ABI (csrc/comm.cpp: kc_comm_init, kc_live_graph_exchange, kc_live_graph_evaluate_partitioned) in a
world of ONE rank whose transfers go from rank 0 to rank 0, over both wires (RCCL allows a send to self inside a group; the IPC
wire reads its own planes directly).  Every part of the path runs -- the slot description through the mailbox, plane data
behind the compute stream, constant planes as scalars, aliased planes once, caller-owned planes through a dense copy, the
import on the receiving side -- and what comes back must be the oracle's result bit for bit.  Several ranks:
tests/test_gpu_comm_ranks.py (processes sharing the GPU); the plans themselves on CPU: tests/test_multi_gpu_gloo.py."""
import json

import numpy as np
import pytest

from test_gpu_partitioned import _device_graph, _oracle
from test_multi_gpu_gloo import diamond_fanin_broadcast_graph, fanin_graph
from util import SEED_A, SEED_B, assert_planes, bit_equal, splitmix_plane

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["ipc", "rccl"])
def kc(request):
    """Both wires: the IPC one reads its own planes directly when a rank sends to itself, RCCL sends to self inside a group."""
    import os
    import kanter_core_amd as kc
    kc.init(0)
    os.environ["KC_COMM_TRANSPORT"] = request.param
    try:
        kc.comm_init(0, 1, kc.comm_unique_id())
    finally:
        del os.environ["KC_COMM_TRANSPORT"]
    assert kc.comm_info() == (0, 1) and kc.comm_transport() == request.param
    yield kc
    kc.comm_destroy()
    assert kc.comm_info() == (0, 0) and kc.comm_transport() == ""


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    return orc


def test_self_transfer_of_an_rgba_mix_result(kc, orc):
    """n1 = Mix(Add)(A, B) is 'sent' (three planes; its constant alpha inside the description), received into fresh planes
    and imported; n2 = Mix(Multiply)(n1, B) then consumes the received copy."""
    h, w = 96, 200
    a = [splitmix_plane(SEED_A, c, h, w) for c in range(4)]
    b = [splitmix_plane(SEED_B, c, h, w) for c in range(4)]
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    na = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
    nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 0)
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(b)), 1)
    n1 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
    n2 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply)))
    lg.connect(na, n1, 0, 0)
    lg.connect(nb, n1, 0, 1)
    lg.connect(n1, n2, 0, 0)
    lg.connect(nb, n2, 0, 1)
    lg.use_cache = True  # n1's slot is kept, so the exchange has something to replace
    s0 = kc.comm_stats()
    lg.exchange([(n1, 0, 0, 0)])
    s1 = kc.comm_stats()
    assert s1["planes_sent"] - s0["planes_sent"] == 3 and s1["planes_received"] - s0["planes_received"] == 3
    assert s1["bytes_sent"] - s0["bytes_sent"] == 3 * h * 1024  # whole pitched buffers: 200 floats -> 1024-byte rows
    got1 = lg.slot_data(n1, 0).image
    want1 = [orc.mix_plane("Add", a[c], b[c]) for c in range(3)] + [np.ones((h, w), np.float32)]
    assert_planes(got1.planes(), want1, what="received slot")
    got2 = lg.await_clean(n2).slot_data(n2, 0).image.planes()
    want2 = [orc.mix_plane("Multiply", want1[c], b[c]) for c in range(3)] + [np.ones((h, w), np.float32)]
    assert_planes(got2, want2, what="consumer of the received slot")
    # a second round replaces the slot again (the first round's planes go back to the pool once their send has fired)
    lg.exchange([(n1, 0, 0, 0)])
    assert_planes(lg.slot_data(n1, 0).image.planes(), want1, what="second round")
    kc.sync()


def test_self_transfer_of_sources_aliased_and_constant_planes(kc, orc):
    """A gray source widened to RGBA is [p, p, p, ones]: one plane travels, the constant goes as a scalar."""
    h, w = 40, 64
    p = splitmix_plane(SEED_A, 7, h, w)
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    src = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes([p]).as_type(True)), 0)
    out = lg.add_node(kc.Node.new(kc.NodeType.OutputRgba("o")))
    lg.connect(src, out, 0, 0)
    s0 = kc.comm_stats()
    lg.exchange([(src, 0, 0, 0)])
    s1 = kc.comm_stats()
    assert s1["planes_sent"] - s0["planes_sent"] == 1 and s1["planes_received"] - s0["planes_received"] == 1
    got = lg.slot_data(src, 0).image
    assert got.is_rgba()
    assert_planes(got.planes(), [p, p, p, np.ones((h, w), np.float32)], what="aliased planes")
    handles = got.plane_handles()
    try:
        assert handles[0] == handles[1] == handles[2] != handles[3]  # still ONE plane behind R, G and B
    finally:
        from kanter_core_amd import _lib
        for hnd in handles:
            _lib.load().kc_plane_release(hnd)


def test_self_transfer_from_caller_owned_memory(kc, orc):
    """A source wrapped around caller memory with a tight pitch is copied to a pool-pitched plane for the wire."""
    import ctypes as C
    import torch
    from kanter_core_amd import _lib
    L = _lib.load()
    h, w, pitch_f = 19, 10, 12
    t = torch.full((h, pitch_f), float("nan"), device="cuda")
    p = splitmix_plane(SEED_B, 3, h, w)
    t[:, :w] = torch.from_numpy(p).cuda()
    torch.cuda.synchronize()
    plane, img = C.c_void_p(), C.c_void_p()
    assert L.kc_plane_wrap(t.data_ptr(), w, h, pitch_f * 4, C.byref(plane)) == 0
    L.kc_image_gray(plane, C.byref(img))
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    src = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage(img.value)), 0)
    lg.exchange([(src, 0, 0, 0)])
    assert bit_equal(lg.slot_data(src, 0).image.planes()[0], p)
    L.kc_plane_release(plane)


@pytest.mark.parametrize("which", ["diamond", "fanin"])
def test_evaluate_partitioned_world1_and_forced_cuts(kc, which):
    """evaluate_partitioned on a one-rank plan (no transfers) is the plain evaluation; then the cut slots of the TWO-rank plan
    of the same graph are pushed through the wire rank 0 -> rank 0 before the root is evaluated: same result."""
    graph, root = (diamond_fanin_broadcast_graph()[:2] if which == "diamond" else fanin_graph(8, 4))
    want = _oracle(graph, root)
    tp, lg = _device_graph(kc, graph)
    plan = lg.partition(root, 1, kc.PartitionPolicy.Spread)
    assert plan.transfers == []
    assert_planes(lg.evaluate_partitioned(plan, root).planes(), want, what=which)
    tp2, lg2 = _device_graph(kc, graph)
    cuts = lg2.partition(root, 2, kc.PartitionPolicy.Spread).transfers
    assert cuts
    s0 = kc.comm_stats()
    lg2.exchange([(n, s, 0, 0) for (n, s, _src, _dst, _lv) in cuts])
    assert kc.comm_stats()["planes_received"] > s0["planes_received"]
    assert_planes(lg2.await_clean(root).slot_data(root, 0).image.planes(), want, what=which + " through the wire")


def test_exchange_without_a_communicator_is_an_error():
    import kanter_core_amd as kc2
    # (module fixture `kc` may or may not be alive here; use a rank outside the communicator instead)
    tp = kc2.TextureProcessor.new()
    lg = tp.new_live_graph()
    n = lg.add_node(kc2.Node.new(kc2.NodeType.Value(0.5)))
    if kc2.comm_info() == (0, 0):
        with pytest.raises(kc2.TexProError):
            lg.exchange([(n, 0, 0, 0)])
    else:
        with pytest.raises(kc2.TexProError):
            lg.exchange([(n, 0, 0, 5)])
