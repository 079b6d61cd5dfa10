"""Multi-GPU evaluation of texture graphs: one process per GPU (torch.distributed; backend "nccl" = RCCL over
xGMI on ROCm, "gloo" on CPU for the tests), every process holds the same graph.

The reference has no distributed layer at all (one OS thread per ready node, src/engine.rs:288); what makes one
possible is its readiness rule: a node needs nothing but its parents' slot data (src/engine.rs:213-275).  Two
ways to shard follow:

  * graph level -- `PartitionedEvaluator`: the library's partitioner (csrc/partition.cpp, C ABI
    kc_live_graph_partition) says which rank evaluates which node and which slots cross a rank boundary, and the
    library moves those slots itself (csrc/comm.cpp, C ABI kc_comm_init + kc_live_graph_evaluate_partitioned): RCCL
    send / recv of whole pitched planes on a communication stream behind an event of the compute stream, the slot's
    64-byte description on a second communicator, constants as scalars, one slot with consumers on several ranks
    sent once per consumer.  Over xGMI every producer owns a distinct link into the consumer, so the inbound
    transfers of a fan-in run concurrently, and the hand-off of one branch overlaps the evaluation of the next one
    on the same rank; nobody's host waits for plane data.  With the `nccl` backend this class is a thin caller of
    that path (torch.distributed only carries the communicator's identifier, once).  The host-side loop below is
    what the CPU tests (gloo, a host slot store) and the several-ranks-on-one-GPU rehearsal run.
  * data level -- `row_bands`: pointwise graphs split by rows with no exchange at all; graphs with resize or
    HeightToNormal nodes through kc_live_graph_await_clean_band, which widens each band by the halo rows the
    node types below it need (bands.cpp).

There is no reduction in this workload, hence no all-reduce.
"""
import time

import torch
import torch.distributed as dist

from . import api as kc
from . import _lib

import ctypes as C


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


# ------------------------------------------------------------------------------------------------ backends
class DeviceBackend:
    """Slots of the library's LiveGraph <-> torch tensors that alias the library's HBM planes (no copies: a plane
    is sent and received as its whole pitched buffer, rows * pitch bytes, padding included)."""

    def __init__(self, live_graph, device):
        self.lg, self.device, self.L = live_graph, device, _lib.load()
        # The library enqueues on its own stream, torch on its current one: nothing orders the two unless they are the
        # same stream.  This backend reads planes the library has only enqueued (export) and writes planes the library's
        # queued kernels may still read (recycled pool blocks), so both sides are put on ONE dedicated stream -- never the
        # legacy default stream, which other non-blocking streams do not wait for.
        cur = torch.cuda.current_stream(device)
        self.stream = cur if cur.cuda_stream != 0 else torch.cuda.Stream(device)
        kc.set_stream(self.stream.cuda_stream)

    def _plane_tensor(self, handle, h):
        ptr, pitch = C.c_void_p(), C.c_size_t()
        kc._check(self.L.kc_plane_device_ptr(handle, C.byref(ptr), C.byref(pitch)))

        class _View:
            __cuda_array_interface__ = {"shape": (h * pitch.value // 4,), "typestr": "<f4", "data": (int(ptr.value), False),
                                        "version": 3, "strides": None}

        return torch.as_tensor(_View(), device=self.device)

    def evaluate(self, node_id):
        self.lg.await_clean(node_id)

    def export_slot(self, node_id, slot_id):
        """-> (header, tensors, keep): header is a small picklable description the receiver needs before it can post its
        receives; constant planes (Mix's alpha = 1, broadcast Values) travel inside it, not as 64 MiB of ones."""
        img = self.lg.slot_data(node_id, slot_id).image
        size = img.size()
        handles = img.plane_handles()  # +1 ref each
        planes, tensors, seen = [], [], {}
        try:
            for hnd in handles:
                is_c, v = C.c_int(), C.c_float()
                kc._check(self.L.kc_plane_is_const(hnd, C.byref(is_c), C.byref(v)))
                if is_c.value:
                    planes.append(("c", float(v.value)))
                elif hnd in seen:  # aliased planes (Gray -> Rgba is [p, p, p, ones]) are sent once
                    planes.append(("m", seen[hnd]))
                else:
                    seen[hnd] = len(tensors)
                    planes.append(("m", len(tensors)))
                    tensors.append(self._plane_tensor(hnd, size.height))
        finally:
            for hnd in handles:
                self.L.kc_plane_release(hnd)
        return {"w": size.width, "h": size.height, "planes": planes}, tensors, img

    def alloc_slot(self, header):
        """Fresh library planes for a slot about to be received -> (tensors to receive into, token for import_slot)."""
        n = 1 + max([i for (k, i) in header["planes"] if k == "m"], default=-1)
        handles, tensors = [], []
        for _ in range(n):
            p = C.c_void_p()
            kc._check(self.L.kc_plane_alloc(header["w"], header["h"], C.byref(p)))
            handles.append(p)
            tensors.append(self._plane_tensor(p, header["h"]))
        return tensors, handles

    def import_slot(self, node_id, slot_id, header, handles):
        L = self.L
        planes = []
        for kind, v in header["planes"]:
            if kind == "c":
                p = C.c_void_p()
                kc._check(L.kc_plane_const(header["w"], header["h"], v, C.byref(p)))
                planes.append((p, True))
            else:
                planes.append((handles[v], False))
        im = C.c_void_p()
        if len(planes) == 1:
            kc._check(L.kc_image_gray(planes[0][0], C.byref(im)))
        else:
            arr = (C.c_void_p * 4)(*[p.value for p, _ in planes])
            kc._check(L.kc_image_rgba(arr, C.byref(im)))
        for p, own in planes:
            if own:
                L.kc_plane_release(p)
        for p in handles:
            L.kc_plane_release(p)
        self.lg.import_slot_data(node_id, slot_id, kc.SlotImage(im.value))

    def result(self, node_id, slot_id=0):
        return self.lg.slot_data(node_id, slot_id).image


# ------------------------------------------------------------------------------------------------ the loop
class PartitionedEvaluator:
    """Evaluates `root` of `live_graph` over the ranks of `group`.

        ev = PartitionedEvaluator(lg, root, policy=kc.PartitionPolicy.Spread)
        img = ev.evaluate()        # SlotImage on the home rank, None elsewhere

    Every rank builds the same graph; a rank only has to embed the data of the SOURCE nodes the plan places on it
    (`ev.plan.nodes`).  `backend` adapts the slot store (DeviceBackend by default; the CPU tests plug in a host-side store);
    it may be a factory called with (plan, rank)."""

    def __init__(self, live_graph, root, policy=kc.PartitionPolicy.Spread, group=None, backend=None, device=None,
                 header_group=None):
        self.lg, self.root, self.group = live_graph, root, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.plan = live_graph.partition(root, self.world, policy)
        if backend is None:
            backend = DeviceBackend(live_graph, device)
        elif not hasattr(backend, "evaluate"):
            backend = backend(self.plan, self.rank)  # a factory: the slot store may depend on the placement
        self.backend = backend
        # headers (a few dozen bytes) go over a host-side group so that posting a receive never waits for a GPU
        self.stage_through_host = dist.is_initialized() and dist.get_backend(group) == "gloo"
        # the native path: RCCL inside the library.  Only with the library as slot store and one rank per GPU.
        self.native = isinstance(backend, DeviceBackend) and dist.is_initialized() and dist.get_backend(group) == "nccl"
        if self.native and kc.comm_info() == (0, 0):
            box = [kc.comm_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=self._global(0), group=group)
            kc.comm_init(self.rank, self.world, box[0])
        if header_group is None and dist.is_initialized() and dist.get_backend(group) == "nccl" and not self.native:
            header_group = dist.new_group(backend="gloo")  # pickled headers over nccl would go through GPU tensors and host syncs
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
        be, rank = self.backend, self.rank
        if self.native:
            t_start = time.perf_counter()
            s0 = kc.comm_stats()
            result = self.lg.evaluate_partitioned(self.plan, self.root)
            s1 = kc.comm_stats()
            self.stats = {"rank": rank, "host_total_s": time.perf_counter() - t_start, "native": True,
                          "planes_sent": s1["planes_sent"] - s0["planes_sent"],
                          "planes_received": s1["planes_received"] - s0["planes_received"],
                          "bytes_sent": s1["bytes_sent"] - s0["bytes_sent"]}
            return result
        import contextlib
        with (torch.cuda.stream(be.stream) if getattr(be, "stream", None) is not None else contextlib.nullcontext()):
            return self._evaluate_host_loop()

    def _evaluate_host_loop(self):
        be, rank = self.backend, self.rank
        t_start = time.perf_counter()
        t_compute = t_exchange = 0.0
        inflight, keep, n_sent, n_recv, bytes_moved = [], [], 0, 0, 0
        evaluated = set()
        for (node, slot, src, dsts, _level) in self.transfers_by_slot():
            if rank == src:
                t0 = time.perf_counter()
                if node not in evaluated:
                    be.evaluate(node)  # enqueues this branch's kernels; returns without waiting for them
                    evaluated.add(node)
                header, tensors, owner = be.export_slot(node, slot)
                t1 = time.perf_counter()
                t_compute += t1 - t0
                for d in dsts:
                    dist.send_object_list([header], dst=self._global(d), group=self.header_group)
                send = [t.cpu() for t in tensors] if self.stage_through_host and tensors and tensors[0].is_cuda else tensors
                ops = [dist.P2POp(dist.isend, t, self._global(d), self.group) for d in dsts for t in send]
                if ops:
                    inflight += dist.batch_isend_irecv(ops)
                keep.append((owner, tensors, send))  # planes stay alive until the sends have completed
                n_sent += len(ops)
                bytes_moved += sum(t.numel() * 4 for t in send) * len(dsts)
                t_exchange += time.perf_counter() - t1
            elif rank in dsts:
                t1 = time.perf_counter()
                box = [None]
                dist.recv_object_list(box, src=self._global(src), group=self.header_group)
                header = box[0]
                tensors, token = be.alloc_slot(header)
                stage = [torch.empty(t.shape, dtype=t.dtype) for t in tensors] if self.stage_through_host and tensors and tensors[0].is_cuda else tensors
                ops = [dist.P2POp(dist.irecv, t, self._global(src), self.group) for t in stage]
                works = dist.batch_isend_irecv(ops) if ops else []
                for w in works:
                    w.wait()  # RCCL: the compute stream waits for the transfer, the host does not
                if stage is not tensors:
                    for t, s in zip(tensors, stage):
                        t.copy_(s)
                be.import_slot(node, slot, header, token)
                n_recv += len(ops)
                t_exchange += time.perf_counter() - t1
        result = None
        t0 = time.perf_counter()
        if rank == self.plan.home:
            be.evaluate(self.root)
            result = be.result(self.root)
        t_compute += time.perf_counter() - t0
        for w in inflight:
            w.wait()
        self._keep = keep  # released by the next evaluate() (stream order has passed the sends by then)
        self.stats = {"rank": rank, "host_compute_s": t_compute, "host_exchange_s": t_exchange, "planes_sent": n_sent,
                      "planes_received": n_recv, "bytes_sent": bytes_moved, "host_total_s": time.perf_counter() - t_start}
        return result

    def _global(self, group_rank):
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank
