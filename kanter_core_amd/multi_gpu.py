"""Multi-GPU evaluation of texture graphs: one process per GPU, every process holds the same graph.

The reference has no distributed layer at all (one OS thread per ready node, src/engine.rs:288); what makes one
possible is its readiness rule: a node needs nothing but its parents' slot data (src/engine.rs:213-275).  Three
plans follow (csrc/partition.cpp decides, the same on every rank without talking to the others):

  * one GPU    -- nothing moves;
  * branches   -- the library's partitioner says which rank evaluates which node and which slots cross a rank boundary;
  * row bands  -- every rank evaluates its rows of the whole graph (pointwise nodes need no exchange at all, resize /
    HeightToNormal nodes a few halo rows, computed redundantly: csrc/bands.cpp); only the finished bands move.

The LIBRARY moves the data (csrc/comm.cpp, C ABI kc_comm_init + kc_live_graph_evaluate_partitioned): descriptions of slots
through a shared-memory mailbox, planes by IPC copies between the processes' HBM (or RCCL send / recv,
KC_COMM_TRANSPORT=rccl), on communication streams ordered with the compute stream on the device.  `PartitionedEvaluator`
is a thin caller of that path; torch.distributed, when it is initialised, only carries the communicator's identifier, once.

What is left in Python is the host-side loop the CPU tests run with a pluggable slot store (`backend=`, gloo): it walks
the same transfer list with torch.distributed send / recv and exists to test PLANS without a GPU.

There is no reduction in this workload, hence no all-reduce.
"""
import time

import torch.distributed as dist

from . import api as kc


def row_bands(height, world_size, align=1):
    """Row bands [y0, y1) per rank.  `align` keeps band starts on a multiple (e.g. the resize tile height)."""
    rows = (height + align - 1) // align
    out, start = [], 0
    for r in range(world_size):
        n = rows // world_size + (1 if r < rows % world_size else 0)
        y0, y1 = min(start * align, height), min((start + n) * align, height)
        out.append((y0, y1))
        start += n
    return out


def ensure_communicator(rank, world, group=None, comm_id=None):
    """The library's communicator for this process (kc_comm_init), created once.  The identifier comes from `comm_id`
    (made by rank 0 with kc.comm_unique_id() and handed over by the caller) or is broadcast over torch.distributed."""
    have = kc.comm_info()
    if have != (0, 0):
        if have != (rank, world):
            raise RuntimeError("a communicator for rank %d of %d exists already" % have)
        return
    if comm_id is None:
        if world == 1:
            comm_id = kc.comm_unique_id()
        else:
            if not dist.is_initialized():
                raise RuntimeError("pass comm_id= (kc.comm_unique_id() of rank 0) or initialise torch.distributed")
            box = [kc.comm_unique_id() if rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            comm_id = box[0]
    kc.comm_init(rank, world, comm_id)


class PartitionedEvaluator:
    """Evaluates `root` of `live_graph` over the ranks of `group` (or of `rank` / `world` given explicitly).

        ev = PartitionedEvaluator(lg, root, policy=kc.PartitionPolicy.Auto)
        img = ev.evaluate()        # SlotImage on the home rank, None elsewhere

    Every rank builds the same graph.  With a branch plan a rank only has to embed the data of the SOURCE nodes the plan
    places on it (`ev.plan.nodes`); with a band plan the rows `lg.band_source_rows(root, *ev.plan.bands[rank])` names (or
    whole sources).  `backend` replaces the library as slot store (the CPU tests plug in a host-side one; it may be a factory
    called with (plan, rank)) and selects the host-side loop."""

    def __init__(self, live_graph, root, policy=kc.PartitionPolicy.Spread, group=None, backend=None, device=None,
                 header_group=None, rank=None, world=None, comm_id=None):
        self.lg, self.root, self.group = live_graph, root, group
        use_dist = dist.is_initialized() and world is None
        self.world = dist.get_world_size(group) if use_dist else (world or 1)
        self.rank = dist.get_rank(group) if use_dist else (rank or 0)
        self.plan = live_graph.partition(root, self.world, policy)
        self.native = backend is None
        if self.native:
            if self.world > 1 or comm_id is not None:
                ensure_communicator(self.rank, self.world, group, comm_id)
        elif not hasattr(backend, "evaluate"):
            backend = backend(self.plan, self.rank)  # a factory: the slot store may depend on the placement
        self.backend = backend
        self.header_group = header_group if header_group is not None else group
        self.stats = {}

    def transfers_by_slot(self):
        """[(node, slot, src, [dst...], level)]: consecutive plan entries of one slot = one (multi-destination) send."""
        out = []
        for (n, s, src, dst, lv) in self.plan.transfers:
            if out and out[-1][0] == n and out[-1][1] == s and out[-1][2] == src:
                out[-1][3].append(dst)
            else:
                out.append((n, s, src, [dst], lv))
        return out

    def evaluate(self):
        if not self.native:
            return self._evaluate_host_loop()
        t_start = time.perf_counter()
        s0 = kc.comm_stats()
        result = self.lg.evaluate_partitioned(self.plan, self.root)
        s1 = kc.comm_stats()
        self.stats = {"rank": self.rank, "host_total_s": time.perf_counter() - t_start, "native": True,
                      "transport": kc.comm_transport(), "plan": ("single", "branches", "bands")[self.plan.kind],
                      "planes_sent": s1["planes_sent"] - s0["planes_sent"],
                      "planes_received": s1["planes_received"] - s0["planes_received"],
                      "bytes_sent": s1["bytes_sent"] - s0["bytes_sent"]}
        return result

    def _evaluate_host_loop(self):
        """Branch plans over torch.distributed with a pluggable slot store (CPU tests)."""
        be, rank = self.backend, self.rank
        if self.plan.kind == kc.PlanKind.Bands:
            raise RuntimeError("the host-side loop walks branch plans only")
        t_start = time.perf_counter()
        inflight, keep, n_sent, n_recv, bytes_moved = [], [], 0, 0, 0
        evaluated = set()
        for (node, slot, src, dsts, _level) in self.transfers_by_slot():
            if rank == src:
                if node not in evaluated:
                    be.evaluate(node)
                    evaluated.add(node)
                header, tensors, owner = be.export_slot(node, slot)
                for d in dsts:
                    dist.send_object_list([header], dst=self._global(d), group=self.header_group)
                ops = [dist.P2POp(dist.isend, t, self._global(d), self.group) for d in dsts for t in tensors]
                if ops:
                    inflight += dist.batch_isend_irecv(ops)
                keep.append((owner, tensors))  # planes stay alive until the sends have completed
                n_sent += len(ops)
                bytes_moved += sum(t.numel() * 4 for t in tensors) * len(dsts)
            elif rank in dsts:
                box = [None]
                dist.recv_object_list(box, src=self._global(src), group=self.header_group)
                header = box[0]
                tensors, token = be.alloc_slot(header)
                ops = [dist.P2POp(dist.irecv, t, self._global(src), self.group) for t in tensors]
                for w in (dist.batch_isend_irecv(ops) if ops else []):
                    w.wait()
                be.import_slot(node, slot, header, token)
                n_recv += len(ops)
        result = None
        if rank == self.plan.home:
            be.evaluate(self.root)
            result = be.result(self.root)
        for w in inflight:
            w.wait()
        self._keep = keep
        self.stats = {"rank": rank, "native": False, "planes_sent": n_sent, "planes_received": n_recv, "bytes_sent": bytes_moved,
                      "host_total_s": time.perf_counter() - t_start}
        return result

    def _global(self, group_rank):
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank
