"""Multi-GPU placement for texture graphs: one process per GPU (torch.distributed, backend "nccl"
= RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no distributed layer at all (one OS thread per ready node, src/engine.rs:288);
what shards naturally is:
  * independent graph branches -> one branch (or several) per GPU, whole planes exchanged only
    where a consumer sits on another GPU (the fan-in of BASELINE config #4);
  * pointwise graphs -> row bands of every plane, with no exchange at all (config #3).
There is no reduction in this workload, hence no all-reduce: the only collective is a gather of
result planes to the consumer's rank.
"""
import torch
import torch.distributed as dist


def assign_branches(n_branches, world_size):
    """Contiguous blocks of independent branches per rank; the first `n % world` ranks take one more."""
    base, extra = divmod(n_branches, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append(list(range(start, start + n)))
        start += n
    return out


def row_bands(height, world_size, align=1):
    """Row bands [y0, y1) per rank for pointwise graphs (Mix / as_type / fill / to_u8 need no halo).
    `align` keeps band starts on a multiple (e.g. the resize tile height)."""
    rows = (height + align - 1) // align
    out, start = [], 0
    for r in range(world_size):
        n = rows // world_size + (1 if r < rows % world_size else 0)
        y0, y1 = min(start * align, height), min((start + n) * align, height)
        out.append((y0, y1))
        start += n
    return out


def gather_planes(planes, dst=0, group=None):
    """Gathers every rank's result planes (list of equally shaped tensors, already on the rank's
    device) to `dst`.  Returns [rank][plane] tensors on dst, None elsewhere.  One gather per plane
    so a plane can leave as soon as it is final; over xGMI each producer owns a distinct link into
    dst, so the 7 inbound transfers of an 8-GPU fan-in run concurrently."""
    if not dist.is_initialized():
        return [list(planes)]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [list(planes)]
    received = [[None] * len(planes) for _ in range(world)] if rank == dst else None
    for i, p in enumerate(planes):
        p = p.contiguous()
        bufs = [torch.empty_like(p) for _ in range(world)] if rank == dst else None
        dist.gather(p, bufs, dst=dst, group=group)
        if rank == dst:
            for r in range(world):
                received[r][i] = bufs[r]
    return received


def fan_in(items, combine):
    """Pairwise reduction tree ((0,1),(2,3)),... of `items` with `combine(a, b)`; the pairing
    order is fixed so results do not depend on arrival order."""
    items = list(items)
    if not items:
        raise ValueError("fan_in of nothing")
    while len(items) > 1:
        nxt = [combine(items[i], items[i + 1]) for i in range(0, len(items) - 1, 2)]
        if len(items) & 1:
            nxt.append(items[-1])
        items = nxt
    return items[0]


class DevicePlaneView:
    """Zero-copy torch view of a library-owned plane (dense planes only: pitch == 4 * width)."""

    def __init__(self, ptr, width, height, pitch):
        if pitch != 4 * width:
            raise ValueError("plane is pitched; gather it row by row or use a width that is a multiple of 64")
        self.__cuda_array_interface__ = {"shape": (height, width), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 3, "strides": None}

    def tensor(self, device):
        return torch.as_tensor(self, device=device)
