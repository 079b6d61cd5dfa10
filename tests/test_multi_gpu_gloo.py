"""The N > 1 path on CPU: world_size-2 (and 3) `gloo` processes exercise branch assignment, row
bands, the plane gather and the fixed-order fan-in tree that bench.py --workload fanin runs over
RCCL.  (Pixel work itself needs the GPU; here the combine callback is a plain tensor add so only
the placement / exchange logic is under test.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kanter_core_amd.multi_gpu import assign_branches, fan_in, gather_planes, row_bands


def test_assign_branches_partitions_everything():
    assert assign_branches(8, 8) == [[i] for i in range(8)]
    assert assign_branches(8, 2) == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert assign_branches(8, 3) == [[0, 1, 2], [3, 4, 5], [6, 7]]
    assert assign_branches(2, 4) == [[0], [1], [], []]
    for n in range(0, 20):
        for w in range(1, 9):
            flat = [b for r in assign_branches(n, w) for b in r]
            assert flat == list(range(n))


def test_row_bands_cover_the_plane():
    assert row_bands(8192, 2) == [(0, 4096), (4096, 8192)]
    assert row_bands(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert row_bands(100, 3, align=16) == [(0, 48), (48, 80), (80, 100)]
    for h in (1, 7, 4096, 8191):
        for w in (1, 2, 3, 8):
            bands = row_bands(h, w)
            assert bands[0][0] == 0 and bands[-1][1] == h
            assert all(bands[i][1] == bands[i + 1][0] for i in range(w - 1))


def test_fan_in_order_is_fixed():
    order = []

    def combine(a, b):
        order.append((a, b))
        return "(%s+%s)" % (a, b)

    assert fan_in(list("abcdefgh"), combine) == "(((a+b)+(c+d))+((e+f)+(g+h)))"
    assert fan_in(list("abc"), combine) == "((a+b)+c)"
    assert fan_in(["x"], combine) == "x"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_branches, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = assign_branches(n_branches, world)[rank]
        # each branch's "result" = 3 planes whose values identify (branch, plane)
        results = [[torch.full((4, 6), float(100 * b + c)) for c in range(3)] for b in mine]
        # ranks may own different numbers of branches: pad to the max so every rank joins every gather
        most = max(len(x) for x in assign_branches(n_branches, world))
        gathered_branches = []
        for slot in range(most):
            planes = results[slot] if slot < len(mine) else [torch.zeros(4, 6) for _ in range(3)]
            got = gather_planes(planes, dst=0)
            if rank == 0:
                for r in range(world):
                    owned = assign_branches(n_branches, world)[r]
                    if slot < len(owned):
                        gathered_branches.append((owned[slot], got[r]))
        if rank == 0:
            gathered_branches.sort(key=lambda t: t[0])
            total = fan_in([p for _, p in gathered_branches], lambda a, b: [x + y for x, y in zip(a, b)])
            q.put([float(t[0, 0]) for t in total])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_branches", [(2, 8), (3, 8), (2, 3)])
def test_gather_and_fan_in_gloo(world, n_branches):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_branches, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out == [float(sum(100 * b + c for b in range(n_branches))) for c in range(3)]
