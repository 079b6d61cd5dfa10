"""Host-side mirror of the reference's public API for the hot path -- same names, argument
meaning and error behaviour as vismut_core 0.10.0 -- on top of the C ABI
(include/kanter_core_amd.h).  All pixel work happens in the HIP library; this file only moves
handles around.

    TextureProcessor   src/texture_processor.rs:18-115
    LiveGraph          src/live_graph.rs:63-645
    NodeGraph          src/node_graph.rs:16-590
    Node / NodeType    src/node/mod.rs:113-194, src/node/node_type.rs:13-28
    MixType            src/node/mix.rs:20-27
    ResizePolicy / ResizeFilter   src/node/mod.rs:33-99
    SlotImage / SlotData / Size   src/slot_image.rs, src/slot_data.rs
    TexProError        src/error.rs:5-27
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import kc_edge, kc_node_desc, kc_size

# ------------------------------------------------------------------ errors (src/error.rs)
_ERROR_NAMES = {
    1: "Generic", 2: "Canceled", 3: "Image", 4: "InvalidBufferCount", 5: "InvalidNodeId", 6: "InvalidNodeType",
    7: "InvalidSlotId", 8: "InvalidSlotType", 9: "InvalidEdge", 10: "NoSlotData", 11: "SlotOccupied",
    12: "SlotNotOccupied", 13: "UnableToLock", 14: "NodeProcessing", 15: "PoisonError", 16: "TryLockError",
    17: "NodeDirty", 18: "Io", 19: "InvalidName", 100: "Hip", 101: "NoDevice", 102: "InvalidArgument",
    103: "OutOfMemory", 104: "Unsupported",
}


class TexProError(Exception):
    """TexProError; `.kind` is the variant name, `.code` the C-ABI status."""

    def __init__(self, code, detail=""):
        self.code = code
        self.kind = _ERROR_NAMES.get(code, "Unknown")
        L = _lib.load()
        msg = L.kc_status_string(code).decode()
        super().__init__("%s: %s%s" % (self.kind, msg, (" (" + detail + ")") if detail else ""))

    def __eq__(self, other):  # discriminant-only PartialEq, src/error.rs:29-33
        return isinstance(other, TexProError) and other.code == self.code

    __hash__ = Exception.__hash__


def _check(status):
    if status != 0:
        L = _lib.load()
        detail = L.kc_last_error() or b""
        raise TexProError(status, detail.decode(errors="replace") if status >= 100 or status in (1, 3, 14, 18) else "")
    return status


def init(device=None):
    """Binds this process to one MI355X (one process per GPU).  Device defaults to LOCAL_RANK."""
    L = _lib.load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    _check(L.kc_init(int(device)))


def is_initialized():
    return bool(_lib.load().kc_is_initialized())


def shutdown():
    _check(_lib.load().kc_shutdown())


def sync():
    _check(_lib.load().kc_sync())


def set_stream(hip_stream):
    """Run on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream); None = own."""
    _check(_lib.load().kc_set_stream(C.c_void_p(hip_stream) if hip_stream else None))


def get_stream():
    return _lib.load().kc_get_stream()


def set_fusion(enabled):
    _check(_lib.load().kc_set_fusion(int(bool(enabled))))


def set_resize_mode(mode):
    """0 all resize kernels, 1-3 A/B switches of the down-sampling kernels, 4 no integer-ratio up-sampling kernels."""
    _check(_lib.load().kc_set_resize_mode(int(mode)))


def get_resize_mode():
    return _lib.load().kc_get_resize_mode()


def set_cache_policy(mode):
    """1: launches that stream more than the Infinity Cache holds mark those streams nontemporal; 0: plain accesses."""
    _check(_lib.load().kc_set_cache_policy(int(mode)))


def get_cache_policy():
    return _lib.load().kc_get_cache_policy()


COMM_ID_BYTES = 256


def comm_unique_id():
    """Rank 0: the identifier every rank passes to comm_init (bytes; hand it over by any host-side channel)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(_lib.load().kc_comm_unique_id(buf))
    return buf.raw


def comm_init(rank, world_size, unique_id):
    """Collective: this process becomes `rank` of `world_size` in the library's communicator (after init()): a shared-memory
    mailbox plus the wire rank 0 chose when it made the id ("ipc" by default, KC_COMM_TRANSPORT=rccl for RCCL)."""
    assert len(unique_id) == COMM_ID_BYTES
    _check(_lib.load().kc_comm_init(int(rank), int(world_size), C.create_string_buffer(unique_id, COMM_ID_BYTES)))


def comm_destroy():
    _check(_lib.load().kc_comm_destroy())


def comm_info():
    r, w = C.c_int(), C.c_int()
    _check(_lib.load().kc_comm_info(C.byref(r), C.byref(w)))
    return r.value, w.value


def comm_transport():
    """"ipc", "rccl" or "" (no communicator)."""
    buf = C.create_string_buffer(16)
    _check(_lib.load().kc_comm_transport(buf, len(buf)))
    return buf.value.decode()


def comm_gather_bands(band, y0, full_height, home_rank=0):
    """Every rank passes its band (a SlotImage holding rows y0 .. of a `full_height`-row image); the assembled SlotImage on
    `home_rank`, None elsewhere (csrc/comm.cpp)."""
    out = C.c_void_p()
    _check(_lib.load().kc_comm_gather_bands(band._h, int(y0), int(full_height), int(home_rank), C.byref(out)))
    return SlotImage(out.value) if out.value else None


def comm_stats():
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _check(_lib.load().kc_comm_stats(C.byref(a), C.byref(b), C.byref(c)))
    return {"planes_sent": a.value, "planes_received": b.value, "bytes_sent": c.value}


def set_option(name, value):
    """Named A/B and test switches (kc_set_option), e.g. "chain1"."""
    _check(_lib.load().kc_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int()
    _check(_lib.load().kc_get_option(name.encode(), C.byref(v)))
    return v.value


def resize_upsample_plan(in_n, out_n, filter=None):
    """Host only: the structure the integer-ratio up-sampling kernels rely on for one axis (kc_resize_upsample_plan),
    as a dict {ratio, taps, off, b_lo, b_hi, rows: float32[ratio + b_lo + b_hi, taps]}, or None when the axis
    does not have it."""
    import numpy as np
    L = _lib.load()
    f = ResizeFilter.default() if filter is None else filter
    ok = C.c_int(0)
    info = (C.c_int32 * 5)()
    _check(L.kc_resize_upsample_plan(int(in_n), int(out_n), int(f), C.byref(ok), info, None, 0))
    if not ok.value:
        return None
    ratio, taps, off, b_lo, b_hi = (int(v) for v in info)
    rows = np.zeros(((ratio + b_lo + b_hi), taps), np.float32)
    _check(L.kc_resize_upsample_plan(int(in_n), int(out_n), int(f), C.byref(ok), info,
                                     rows.ctypes.data_as(C.POINTER(C.c_float)), rows.size))
    return dict(ratio=ratio, taps=taps, off=off, b_lo=b_lo, b_hi=b_hi, rows=rows)


def resize_down2_plan(in_n, out_n, filter=None):
    """Host only: the tables resize_down2_kernel reads for one axis (kc_resize_down2_plan), next to the plain tap table they
    are built from: dict(stride, nc, hstride, tile_w, min_count, left, count, w[out_n, stride], vrec[groups, nc, 72] (uint32),
    hw[out_n, hstride])."""
    import numpy as np
    L = _lib.load()
    f = ResizeFilter.default() if filter is None else filter
    info = (C.c_int32 * 5)()
    _check(L.kc_resize_down2_plan(int(in_n), int(out_n), int(f), info, None, None, 0, None, 0, None, 0))
    stride, nc, hstride, tile_w, min_count = (int(v) for v in info)
    groups = (out_n + 3) // 4
    lc = np.zeros(2 * out_n, np.uint32)
    w = np.zeros((out_n, stride), np.float32)
    vrec = np.zeros((groups, max(nc, 1), 72), np.uint32)
    hw = np.zeros((out_n, max(hstride, 1)), np.float32)
    _check(L.kc_resize_down2_plan(int(in_n), int(out_n), int(f), info, lc.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  w.ctypes.data_as(C.POINTER(C.c_float)), w.size,
                                  vrec.ctypes.data_as(C.POINTER(C.c_uint32)), vrec.size if nc else 0,
                                  hw.ctypes.data_as(C.POINTER(C.c_float)), hw.size if hstride else 0))
    return dict(stride=stride, nc=nc, hstride=hstride, tile_w=tile_w, min_count=min_count, left=lc[:out_n].copy(),
                count=lc[out_n:].copy(), w=w, vrec=vrec if nc else None, hw=hw if hstride else None)


def set_specialize(mode, after=0):
    """Run-time specialisation of the fused chain kernel: 0 interpreter only, 1 compile in the background once a
    program has been seen `after` times (default), 2 compile at first sight and wait.  Bit-identical results."""
    _check(_lib.load().kc_set_specialize(int(mode), int(after)))


def get_specialize():
    return _lib.load().kc_get_specialize()


def specialize_wait():
    """Blocks until every queued kernel compile has landed."""
    _check(_lib.load().kc_specialize_wait())


def specialize_stats():
    v = [C.c_uint64() for _ in range(4)]
    _check(_lib.load().kc_specialize_stats(*[C.byref(x) for x in v]))
    return dict(zip(("kernels_compiled", "compiles_failed", "specialized_launches", "compiles_pending"), [x.value for x in v]))


def specialize_compile_check(words, n_in, start_src=0, flat=True):
    """Generates and compiles (without loading; no device needed) the specialised kernel of a step program.
    Returns the generated source."""
    arr = (C.c_uint32 * len(words))(*words)
    buf = C.create_string_buffer(1 << 16)
    _check(_lib.load().kc_specialize_compile_check(arr, len(words), int(n_in), int(start_src), int(bool(flat)), buf, len(buf)))
    return buf.value.decode()


def specialize_compile_check_upsample(words, n_in, start_src=0, taps=3, wide=True):
    """The same for a program that runs inside the integer-ratio up-sampling kernel (input n_in - 1 = the resampled
    operand).  Returns the generated source."""
    arr = (C.c_uint32 * len(words))(*words)
    buf = C.create_string_buffer(1 << 17)
    _check(_lib.load().kc_specialize_compile_check_upsample(arr, len(words), int(n_in), int(start_src), int(taps), int(bool(wide)),
                                                            buf, len(buf)))
    return buf.value.decode()


def stats():
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _check(_lib.load().kc_stats(C.byref(a), C.byref(b), C.byref(c)))
    d = C.c_uint64()
    _check(_lib.load().kc_stats_algorithmic_bytes(C.byref(d)))
    return {"bytes_in_use": a.value, "bytes_cached": b.value, "kernel_launches": c.value, "algorithmic_bytes": d.value}


def kernel_cache_set_dir(path):
    """Where compiled kernels are kept between processes (csrc/specialize.cpp); None: the environment's choice, "off": nowhere."""
    _check(_lib.load().kc_kernel_cache_set_dir(None if path is None else str(path).encode()))


def kernel_cache_stats():
    v = [C.c_uint64() for _ in range(4)]
    _check(_lib.load().kc_kernel_cache_stats(*[C.byref(x) for x in v]))
    return dict(zip(("files_accepted", "files_refused", "files_written", "kernels_loaded"), (x.value for x in v)))


def kernel_cache_precompile(words, n_in, start_src, flat, nt_mask, directory, up_taps=0, up_wide=False):
    """Compiles the program given by its step words with hiprtc (no device needed) and writes its code object into `directory`."""
    arr = (C.c_uint32 * len(words))(*words)
    _check(_lib.load().kc_kernel_cache_precompile(arr, len(words), int(n_in), int(start_src), int(bool(flat)), int(nt_mask),
                                                  int(up_taps), int(bool(up_wide)), str(directory).encode()))


def specialize_reset():
    """Forgets every kernel this process has compiled or loaded: the next sighting of a program is a first one."""
    _check(_lib.load().kc_specialize_reset())


def pool_trim():
    """kc_pool_trim: waits for the stream and gives every cached (free) block back to the driver."""
    _check(_lib.load().kc_pool_trim())


def stats_counter(name):
    """Named event counter (kc_stats_counter), e.g. "upsample_launches"."""
    v = C.c_uint64()
    _check(_lib.load().kc_stats_counter(name.encode(), C.byref(v)))
    return v.value


# ------------------------------------------------------------------ small value types
class NodeId(int):
    """NodeId(u32), src/node_graph.rs:592-607"""


class SlotId(int):
    """SlotId(u32), src/node_graph.rs:609-624"""


class EmbeddedSlotDataId(int):
    """src/node/embed.rs:13-14"""


class Size:
    """src/slot_data.rs:4-30"""

    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)

    def pixel_count(self):
        return (self.width * self.height) & 0xFFFFFFFF

    def __eq__(self, o):
        if isinstance(o, tuple):
            return (self.width, self.height) == o
        return isinstance(o, Size) and (self.width, self.height) == (o.width, o.height)

    def __hash__(self):
        return hash((self.width, self.height))

    def __iter__(self):
        return iter((self.width, self.height))

    def __repr__(self):
        return "%dx%d" % (self.width, self.height)

    def _c(self):
        return kc_size(self.width, self.height)


class _Enum:
    _names = ()

    @classmethod
    def name_of(cls, v):
        return cls._names[v]

    @classmethod
    def parse(cls, s):
        return cls._names.index(s)


class MixType(_Enum):
    Add, Subtract, Multiply, Divide, Pow = range(5)
    _names = ("Add", "Subtract", "Multiply", "Divide", "Pow")

    @staticmethod
    def default():
        return MixType.Add


class ResizeFilter(_Enum):
    Nearest, Triangle, CatmullRom, Gaussian, Lanczos3 = range(5)
    _names = ("Nearest", "Triangle", "CatmullRom", "Gaussian", "Lanczos3")

    @staticmethod
    def default():
        return ResizeFilter.Triangle


class ResizePolicy:
    """src/node/mod.rs:33-47.  Unit variants are class attributes; SpecificSlot / SpecificSize
    take their payload."""

    def __init__(self, kind, slot=0, size=None):
        self.kind, self.slot, self.size = kind, int(slot), size or Size(0, 0)

    @staticmethod
    def SpecificSlot(slot_id):
        return ResizePolicy(4, slot=slot_id)

    @staticmethod
    def SpecificSize(size):
        return ResizePolicy(5, size=size if isinstance(size, Size) else Size(*size))

    @staticmethod
    def default():
        return ResizePolicy.MostPixels

    def __eq__(self, o):
        return isinstance(o, ResizePolicy) and (self.kind, self.slot, self.size) == (o.kind, o.slot, o.size)

    def __hash__(self):
        return hash((self.kind, self.slot, self.size))


ResizePolicy.MostPixels = ResizePolicy(0)
ResizePolicy.LeastPixels = ResizePolicy(1)
ResizePolicy.LargestAxes = ResizePolicy(2)
ResizePolicy.SmallestAxes = ResizePolicy(3)


class Side:
    Input, Output = 0, 1


class NodeState:
    """src/live_graph.rs:22-37"""
    Clean, Dirty, Requested, Prioritised, Processing, ProcessingDirty = range(6)


class NodeType:
    """src/node/node_type.rs:13-28.  NodeType.Mix(MixType.Add), NodeType.Value(0.5),
    NodeType.Image(path), NodeType.SeparateRgba ..."""
    _KINDS = ("InputGray", "InputRgba", "OutputGray", "OutputRgba", "Graph", "Image", "Embed", "Write", "Value",
              "Mix", "HeightToNormal", "SeparateRgba", "CombineRgba")

    def __init__(self, kind, payload=None):
        self.kind, self.payload = kind, payload

    def __eq__(self, o):  # discriminant-only PartialEq, node_type.rs:50-54
        return isinstance(o, NodeType) and o.kind == self.kind

    def __hash__(self):
        return hash(self.kind)

    def __repr__(self):
        return "%s(%r)" % (self._KINDS[self.kind], self.payload) if self.payload is not None else self._KINDS[self.kind]

    @staticmethod
    def InputGray(name):
        return NodeType(0, str(name))

    @staticmethod
    def InputRgba(name):
        return NodeType(1, str(name))

    @staticmethod
    def OutputGray(name):
        return NodeType(2, str(name))

    @staticmethod
    def OutputRgba(name):
        return NodeType(3, str(name))

    @staticmethod
    def Graph(node_graph):
        return NodeType(4, node_graph)

    @staticmethod
    def Image(path):
        return NodeType(5, os.fspath(path))

    @staticmethod
    def Embed(embedded_slot_data_id):
        return NodeType(6, int(embedded_slot_data_id))

    @staticmethod
    def Write(path):
        return NodeType(7, os.fspath(path))

    @staticmethod
    def Value(value):
        return NodeType(8, float(value))

    @staticmethod
    def Mix(mix_type):
        return NodeType(9, int(mix_type))


NodeType.HeightToNormal = NodeType(10)
NodeType.SeparateRgba = NodeType(11)
NodeType.CombineRgba = NodeType(12)


class Node:
    """src/node/mod.rs:113-194 (priority / cancel are editor scheduling state and not carried)."""

    def __init__(self, node_type, node_id=0):
        self.node_id = NodeId(node_id)
        self.node_type = node_type
        self.resize_policy = ResizePolicy.default()
        self.resize_filter = ResizeFilter.default()

    new = classmethod(lambda cls, node_type: cls(node_type))

    @classmethod
    def with_id(cls, node_type, node_id):
        return cls(node_type, node_id)

    def with_resize_policy(self, policy):
        self.resize_policy = policy
        return self

    def with_resize_filter(self, filt):
        self.resize_filter = filt
        return self

    def filter_type(self, rf):
        self.resize_filter = rf

    def _desc(self):
        d = kc_node_desc()
        d.node_id = int(self.node_id)
        d.node_type = self.node_type.kind
        d.resize_policy = self.resize_policy.kind
        d.policy_slot = self.resize_policy.slot
        d.policy_size = self.resize_policy.size._c()
        d.resize_filter = int(self.resize_filter)
        p = self.node_type.payload
        k = self.node_type.kind
        if k in (0, 1, 2, 3, 5, 7):
            d.text = p.encode()
        elif k == 4:
            d.graph = p._h.value
        elif k == 6:
            d.embed_id = p
        elif k == 8:
            d.value = p
        elif k == 9:
            d.mix_type = p
        return d


class Edge:
    """src/edge.rs:8-74"""

    def __init__(self, output_id, input_id, output_slot, input_slot):
        self.output_id, self.input_id = NodeId(output_id), NodeId(input_id)
        self.output_slot, self.input_slot = SlotId(output_slot), SlotId(input_slot)

    def _c(self):
        return kc_edge(self.output_id, self.input_id, self.output_slot, self.input_slot)

    def __eq__(self, o):
        return isinstance(o, Edge) and self._t() == o._t()

    def _t(self):
        return (self.output_id, self.input_id, self.output_slot, self.input_slot)

    def __hash__(self):
        return hash(self._t())

    def __repr__(self):
        return "Edge(%d:%d -> %d:%d)" % (self.output_id, self.output_slot, self.input_id, self.input_slot)


def _ids(fn, handle, *pre):
    L = _lib.load()
    n = C.c_uint32()
    _check(fn(handle, *pre, None, 0, C.byref(n)))
    arr = (C.c_uint32 * max(n.value, 1))()
    _check(fn(handle, *pre, arr, n.value, C.byref(n)))
    return [arr[i] for i in range(n.value)]


def _edges(fn, handle):
    n = C.c_uint32()
    _check(fn(handle, None, 0, C.byref(n)))
    arr = (kc_edge * max(n.value, 1))()
    _check(fn(handle, arr, n.value, C.byref(n)))
    return [Edge(e.output_id, e.input_id, e.output_slot, e.input_slot) for e in arr[:n.value]]


# ------------------------------------------------------------------ SlotImage / SlotData
class SlotImage:
    """SlotImage (src/slot_image.rs:15-264) backed by device planes."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    def __del__(self):
        try:
            if self._h:
                _lib.load().kc_image_release(self._h)
        except Exception:
            pass

    # constructors
    @staticmethod
    def from_value(size, value, rgba):
        out = C.c_void_p()
        s = size if isinstance(size, Size) else Size(*size)
        _check(_lib.load().kc_image_from_value(s._c(), float(value), int(rgba), C.byref(out)))
        return SlotImage(out.value)

    @staticmethod
    def from_u8(px):
        """deconstruct_image: uint8 array (h, w, channels)."""
        px = np.ascontiguousarray(px, np.uint8)
        if px.ndim == 2:
            px = px[:, :, None]
        h, w, c = px.shape
        out = C.c_void_p()
        _check(_lib.load().kc_image_from_u8(px.ctypes.data, w, h, c, C.byref(out)))
        return SlotImage(out.value)

    @staticmethod
    def from_planes(planes):
        """1 (Gray) or 4 (Rgba) float32 arrays of shape (h, w)."""
        planes = [np.ascontiguousarray(p, np.float32) for p in planes]
        h, w = planes[0].shape
        arr = (C.c_void_p * len(planes))(*[p.ctypes.data for p in planes])
        out = C.c_void_p()
        _check(_lib.load().kc_image_from_f32(arr, len(planes), w, h, C.byref(out)))
        return SlotImage(out.value)

    @staticmethod
    def read_png(path):
        out = C.c_void_p()
        _check(_lib.load().kc_image_read_png(os.fspath(path).encode(), C.byref(out)))
        return SlotImage(out.value)

    # queries
    def is_rgba(self):
        v = C.c_int()
        _check(_lib.load().kc_image_is_rgba(self._h, C.byref(v)))
        return bool(v.value)

    def size(self):
        s = kc_size()
        _check(_lib.load().kc_image_size(self._h, C.byref(s)))
        return Size(s.width, s.height)

    def as_type(self, rgba):
        out = C.c_void_p()
        _check(_lib.load().kc_image_as_type(self._h, int(rgba), C.byref(out)))
        return SlotImage(out.value)

    def to_u8(self, srgb=False):
        """-> uint8 (h, w, 4), src/slot_image.rs:141-170 (srgb: :172-207)"""
        s = self.size()
        out = np.empty((s.height, s.width, 4), np.uint8)
        _check(_lib.load().kc_image_to_u8(self._h, int(srgb), out.ctypes.data))
        return out

    def to_u8_srgb(self):
        return self.to_u8(True)

    def planes(self):
        """Downloads the f32 planes: list of (h, w) arrays (1 or 4)."""
        s = self.size()
        n = 4 if self.is_rgba() else 1
        outs = [np.empty((s.height, s.width), np.float32) for _ in range(n)]
        arr = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        _check(_lib.load().kc_image_to_f32(self._h, arr, n))
        return outs

    def materialize(self):
        _check(_lib.load().kc_image_materialize(self._h))
        return self

    def plane_handles(self):
        L = _lib.load()
        out = []
        for c in range(4 if self.is_rgba() else 1):
            p = C.c_void_p()
            _check(L.kc_image_plane(self._h, c, C.byref(p)))
            out.append(p.value)
        return out

    def write_png(self, path):
        _check(_lib.load().kc_image_write_png(self._h, os.fspath(path).encode()))


class SlotData:
    """src/slot_data.rs:34-79"""

    def __init__(self, node_id, slot_id, image):
        self.node_id, self.slot_id, self.image = NodeId(node_id), SlotId(slot_id), image

    def size(self):
        return self.image.size()

    def in_memory(self):
        return True  # planes live in HBM; there is no disk tier


# ------------------------------------------------------------------ operators (process fns)
def mix_process(left, right, mix_type):
    """mix::process, src/node/mix.rs:51-134; left / right are SlotImage or None."""
    out = C.c_void_p()
    _check(_lib.load().kc_mix_process(left._h if left else None, right._h if right else None, int(mix_type),
                                      C.byref(out)))
    return SlotImage(out.value) if out.value else None


def resize_image(image, size, filt=ResizeFilter.Triangle):
    """image::imageops::resize per plane (src/shared.rs:159-199)."""
    out = C.c_void_p()
    s = size if isinstance(size, Size) else Size(*size)
    _check(_lib.load().kc_resize_image(image._h, s._c(), int(filt), C.byref(out)))
    return SlotImage(out.value)


def height_to_normal_process(image):
    out = C.c_void_p()
    _check(_lib.load().kc_height_to_normal_process(image._h if image else None, C.byref(out)))
    return SlotImage(out.value) if out.value else None


def separate_rgba_process(image):
    outs = (C.c_void_p * 4)()
    _check(_lib.load().kc_separate_rgba_process(image._h if image else None, outs))
    return [SlotImage(o) for o in outs]


def combine_rgba_process(images):
    ins = (C.c_void_p * 4)(*[(i._h.value if i else None) for i in images])
    out = C.c_void_p()
    _check(_lib.load().kc_combine_rgba_process(ins, C.byref(out)))
    return SlotImage(out.value)


def value_process(v):
    out = C.c_void_p()
    _check(_lib.load().kc_value_process(float(v), C.byref(out)))
    return SlotImage(out.value)


def calculate_size(policy, sizes, slot_index=-1):
    """calculate_size, src/shared.rs:61-139; sizes in edge insertion order."""
    arr = (kc_size * max(len(sizes), 1))(*[kc_size(*s) for s in sizes])
    out = kc_size()
    _check(_lib.load().kc_calculate_size(policy.kind, arr, len(sizes), slot_index, policy.size._c(), C.byref(out)))
    return Size(out.width, out.height)


# ------------------------------------------------------------------ NodeGraph
class NodeGraph:
    """src/node_graph.rs:16-590"""

    def __init__(self, handle=None):
        if handle is None:
            h = C.c_void_p()
            _check(_lib.load().kc_node_graph_new(C.byref(h)))
            handle = h.value
        self._h = C.c_void_p(handle)

    new = classmethod(lambda cls: cls())

    def __del__(self):
        try:
            _lib.load().kc_node_graph_free(self._h)
        except Exception:
            pass

    @staticmethod
    def from_path(path):
        h = C.c_void_p()
        _check(_lib.load().kc_node_graph_from_path(os.fspath(path).encode(), C.byref(h)))
        return NodeGraph(h.value)

    @staticmethod
    def from_json(text):
        h = C.c_void_p()
        _check(_lib.load().kc_node_graph_from_json(text.encode(), C.byref(h)))
        return NodeGraph(h.value)

    def to_json(self):
        L = _lib.load()
        n = C.c_size_t()
        _check(L.kc_node_graph_to_json(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(L.kc_node_graph_to_json(self._h, buf, n.value, C.byref(n)))
        return buf.value.decode()

    def export_json(self, path):
        _check(_lib.load().kc_node_graph_export_json(self._h, os.fspath(path).encode()))

    def add_node(self, node):
        nid = C.c_uint32()
        _check(_lib.load().kc_node_graph_add_node(self._h, C.byref(node._desc()), C.byref(nid)))
        return NodeId(nid.value)

    def add_node_with_id(self, node):
        _check(_lib.load().kc_node_graph_add_node_with_id(self._h, C.byref(node._desc())))

    def connect(self, output_node, input_node, output_slot, input_slot):
        _check(_lib.load().kc_node_graph_connect(self._h, output_node, input_node, output_slot, input_slot))
        return Edge(output_node, input_node, output_slot, input_slot)

    def try_connect(self, output_node, input_node, output_slot, input_slot):
        _check(_lib.load().kc_node_graph_try_connect(self._h, output_node, input_node, output_slot, input_slot))

    def remove_node(self, node_id):
        _check(_lib.load().kc_node_graph_remove_node(self._h, node_id))

    def remove_edge(self, edge):
        _check(_lib.load().kc_node_graph_remove_edge(self._h, edge._c()))

    def disconnect_slot(self, node_id, side, slot_id):
        _check(_lib.load().kc_node_graph_disconnect_slot(self._h, node_id, side, slot_id))

    def node_ids(self):
        return [NodeId(i) for i in _ids(_lib.load().kc_node_graph_node_ids, self._h)]

    def edges(self):
        return _edges(_lib.load().kc_node_graph_edges, self._h)

    def input_slot_id_with_name(self, name):
        v = C.c_uint32()
        s = _lib.load().kc_node_graph_input_slot_id_with_name(self._h, name.encode(), C.byref(v))
        return SlotId(v.value) if s == 0 else None

    def output_slot_id_with_name(self, name):
        v = C.c_uint32()
        s = _lib.load().kc_node_graph_output_slot_id_with_name(self._h, name.encode(), C.byref(v))
        return SlotId(v.value) if s == 0 else None

    def set_mix_type(self, node_id, mix_type):
        _check(_lib.load().kc_node_graph_set_mix_type(self._h, node_id, int(mix_type)))

    def set_image_node_path(self, node_id, path):
        _check(_lib.load().kc_node_graph_set_image_node_path(self._h, node_id, os.fspath(path).encode()))

    def rename_output_node(self, node_id, new_name):
        """-> the old name (src/node_graph.rs:232-269)"""
        buf = C.create_string_buffer(1024)
        _check(_lib.load().kc_node_graph_rename_output_node(self._h, node_id, new_name.encode(), buf, 1024))
        return buf.value.decode()


# ------------------------------------------------------------------ LiveGraph / TextureProcessor
class LiveGraph:
    """src/live_graph.rs:63-645.  `await_clean_read` evaluates synchronously on the HIP stream
    (the reference spins on a background scheduler, src/live_graph.rs:181-195)."""

    def __init__(self, handle, tex_pro):
        self._h = C.c_void_p(handle)
        self._tp = tex_pro

    def __del__(self):
        try:
            _lib.load().kc_live_graph_free(self._h)
        except Exception:
            pass

    # RwLock-guard shims so code can read like the reference's tests
    def write(self):
        return self

    def read(self):
        return self

    def unwrap(self):
        return self

    # flags
    def _flags(self):
        a, u = C.c_int(), C.c_int()
        _check(_lib.load().kc_live_graph_get_flags(self._h, C.byref(a), C.byref(u)))
        return bool(a.value), bool(u.value)

    @property
    def auto_update(self):
        return self._flags()[0]

    @auto_update.setter
    def auto_update(self, v):
        _check(_lib.load().kc_live_graph_set_flags(self._h, int(v), int(self._flags()[1])))

    @property
    def use_cache(self):
        return self._flags()[1]

    @use_cache.setter
    def use_cache(self, v):
        _check(_lib.load().kc_live_graph_set_flags(self._h, int(self._flags()[0]), int(v)))

    def set_base_dir(self, path):
        _check(_lib.load().kc_live_graph_set_base_dir(self._h, os.fspath(path).encode()))

    # structure
    def set_node_graph(self, node_graph):
        _check(_lib.load().kc_live_graph_set_node_graph(self._h, node_graph._h))

    def node_graph(self):
        h = C.c_void_p()
        _check(_lib.load().kc_live_graph_node_graph(self._h, C.byref(h)))
        return NodeGraph(h.value)

    def add_node(self, node):
        nid = C.c_uint32()
        _check(_lib.load().kc_live_graph_add_node(self._h, C.byref(node._desc()), C.byref(nid)))
        return NodeId(nid.value)

    def add_node_with_id(self, node):
        _check(_lib.load().kc_live_graph_add_node_with_id(self._h, C.byref(node._desc())))

    def remove_node(self, node_id):
        _check(_lib.load().kc_live_graph_remove_node(self._h, node_id))

    def connect(self, output_node, input_node, output_slot, input_slot):
        _check(_lib.load().kc_live_graph_connect(self._h, output_node, input_node, output_slot, input_slot))
        return Edge(output_node, input_node, output_slot, input_slot)

    def remove_edge(self, edge):
        _check(_lib.load().kc_live_graph_remove_edge(self._h, edge._c()))

    def disconnect_slot(self, node_id, side, slot_id):
        _check(_lib.load().kc_live_graph_disconnect_slot(self._h, node_id, side, slot_id))

    def set_mix_type(self, node_id, mix_type):
        _check(_lib.load().kc_live_graph_set_mix_type(self._h, node_id, int(mix_type)))

    def rename_output_node(self, node_id, new_name):
        buf = C.create_string_buffer(1024)
        _check(_lib.load().kc_live_graph_rename_output_node(self._h, node_id, new_name.encode(), buf, 1024))
        return buf.value.decode()

    def set_resize(self, node_id, policy=None, filt=None):
        policy = policy or ResizePolicy.default()
        filt = ResizeFilter.default() if filt is None else filt
        _check(_lib.load().kc_live_graph_set_resize(self._h, node_id, policy.kind, policy.slot, policy.size._c(), int(filt)))

    def node_ids(self):
        return [NodeId(i) for i in _ids(_lib.load().kc_live_graph_node_ids, self._h)]

    def output_ids(self):
        return [NodeId(i) for i in _ids(_lib.load().kc_live_graph_output_ids, self._h)]

    def edges(self):
        return _edges(_lib.load().kc_live_graph_edges, self._h)

    def changed_consume(self):
        return [NodeId(i) for i in _ids(_lib.load().kc_live_graph_changed_consume, self._h)]

    # state / evaluation
    def node_state(self, node_id):
        v = C.c_int()
        _check(_lib.load().kc_live_graph_node_state(self._h, node_id, C.byref(v)))
        return v.value

    def request(self, node_id):
        _check(_lib.load().kc_live_graph_request(self._h, node_id))

    def prioritise(self, node_id):
        _check(_lib.load().kc_live_graph_prioritise(self._h, node_id))

    def update(self):
        _check(_lib.load().kc_live_graph_update(self._h))

    def await_clean(self, node_id):
        _check(_lib.load().kc_live_graph_await_clean(self._h, node_id))
        return self

    @staticmethod
    def await_clean_read(live_graph, node_id):
        return live_graph.await_clean(node_id)

    await_clean_write = await_clean_read

    # results
    def slot_data(self, node_id, slot_id):
        out = C.c_void_p()
        _check(_lib.load().kc_live_graph_slot_data(self._h, node_id, slot_id, C.byref(out)))
        return SlotData(node_id, slot_id, SlotImage(out.value))

    def node_slot_datas(self, node_id):
        fn = _lib.load().kc_live_graph_node_slot_ids
        return [self.slot_data(node_id, s) for s in _ids(fn, self._h, node_id)]

    def slot_data_size(self, node_id, slot_id):
        s = kc_size()
        _check(_lib.load().kc_live_graph_slot_data_size(self._h, node_id, slot_id, C.byref(s)))
        return Size(s.width, s.height)

    def slot_in_memory(self, node_id, slot_id):
        v = C.c_int()
        _check(_lib.load().kc_live_graph_slot_in_memory(self._h, node_id, slot_id, C.byref(v)))
        return bool(v.value)

    def buffer_rgba(self, node_id, slot_id, srgb=False):
        """-> uint8 (h, w, 4); src/live_graph.rs:93-95"""
        s = self.slot_data_size(node_id, slot_id)
        out = np.empty((s.height, s.width, 4), np.uint8)
        _check(_lib.load().kc_live_graph_buffer_rgba(self._h, node_id, slot_id, int(srgb), out.ctypes.data))
        return out

    @staticmethod
    def try_buffer_rgba(live_graph, node_id, slot_id, srgb=False):
        """src/live_graph.rs:98-124: the buffer when the node is Clean, else the node is requested and
        TexProError(InvalidNodeId) is raised -- the caller polls (update() / await_clean() progresses)."""
        if live_graph.node_state(node_id) == NodeState.Clean:
            return live_graph.buffer_rgba(node_id, slot_id, srgb)
        live_graph.request(node_id)
        raise TexProError(5)

    @staticmethod
    def try_buffer_srgba(live_graph, node_id, slot_id):
        """src/live_graph.rs:126-153"""
        return LiveGraph.try_buffer_rgba(live_graph, node_id, slot_id, srgb=True)

    def embed_slot_data_with_id(self, slot_data, embedded_id):
        _check(_lib.load().kc_live_graph_embed_slot_data_with_id(self._h, slot_data.image._h, slot_data.slot_id,
                                                                 int(embedded_id)))
        return EmbeddedSlotDataId(embedded_id)

    def partition(self, root_node_id, world_size, policy=0):
        """Multi-GPU placement of the evaluation of `root_node_id` over `world_size` ranks (csrc/partition.cpp);
        the same on every rank.  policy: PartitionPolicy.Auto / .Spread / .Bands."""
        h = C.c_void_p()
        _check(_lib.load().kc_live_graph_partition(self._h, int(root_node_id), int(world_size), int(policy), C.byref(h)))
        return Partition(h.value)

    def evaluate_band(self, node_id, y0, y1, slot_id=0):
        """Rows [y0, y1) of the node's result (an image of y1 - y0 rows), bit-identical to those rows of the whole-image
        evaluation; intermediate bands carry the halo rows resize / HeightToNormal nodes need (csrc/bands.cpp)."""
        out = C.c_void_p()
        _check(_lib.load().kc_live_graph_evaluate_band(self._h, int(node_id), int(slot_id), int(y0), int(y1), C.byref(out)))
        return SlotImage(out.value)

    def band_source_rows(self, node_id, y0, y1):
        """{source node id: (y0, y1, width, height)}: the rows of every source image that band reads (y0 < 0: wrapped)."""
        from ._lib import kc_band_rows
        n = C.c_uint32()
        _check(_lib.load().kc_live_graph_band_source_rows(self._h, int(node_id), int(y0), int(y1), None, 0, C.byref(n)))
        buf = (kc_band_rows * max(n.value, 1))()
        _check(_lib.load().kc_live_graph_band_source_rows(self._h, int(node_id), int(y0), int(y1), buf, n.value, C.byref(n)))
        return {buf[i].node_id: (buf[i].y0, buf[i].y1, buf[i].width, buf[i].height) for i in range(n.value)}

    def embed_slot_data_band(self, slot_data, embedded_id, band_y0, full_height):
        """Embeds an image that holds only rows band_y0 .. of a `full_height`-row image (row-band evaluation)."""
        _check(_lib.load().kc_live_graph_embed_slot_data_band(self._h, slot_data.image._h, slot_data.slot_id, int(embedded_id),
                                                              int(band_y0), int(full_height)))
        return EmbeddedSlotDataId(embedded_id)

    def exchange(self, transfers):
        """Moves the slots named by `transfers` = [(node_id, slot_id, src_rank, dst_rank[, level])] between the ranks of the
        communicator (comm_init) with RCCL, inside the library (csrc/comm.cpp); every rank passes the same list."""
        from ._lib import kc_transfer
        n = len(transfers)
        buf = (kc_transfer * max(n, 1))()
        for i, t in enumerate(transfers):
            buf[i].node_id, buf[i].slot_id, buf[i].src_rank, buf[i].dst_rank = int(t[0]), int(t[1]), int(t[2]), int(t[3])
            buf[i].level = int(t[4]) if len(t) > 4 else 0
        _check(_lib.load().kc_live_graph_exchange(self._h, buf, n))

    def evaluate_partitioned(self, plan, root_node_id):
        """The exchange of `plan` (a Partition) followed by `root_node_id` on the plan's home rank: the root's image there,
        None on the other ranks.  A band plan: this rank's rows, gathered on the home rank (or, after plan.set_gather(False),
        the band itself on every rank)."""
        out = C.c_void_p()
        _check(_lib.load().kc_live_graph_evaluate_partitioned(self._h, plan._h, int(root_node_id), C.byref(out)))
        return SlotImage(out.value) if out.value else None

    def import_slot_data(self, node_id, slot_id, image):
        """The receiving side of a transfer: `image` becomes slot `slot_id` of `node_id`, which is Clean afterwards."""
        _check(_lib.load().kc_live_graph_import_slot_data(self._h, int(node_id), int(slot_id), image._h))

    def add_input_slot_data(self, slot_data):
        _check(_lib.load().kc_live_graph_add_input_slot_data(self._h, slot_data.node_id, slot_data.slot_id,
                                                             slot_data.image._h))


class U8Pipe:
    """The u8 host boundary as a pipeline (csrc/u8pipe.cpp; include/kanter_core_amd.h kc_u8_pipe_*): `depth` slots of pinned
    buffers, uploads and downloads on copy streams of their own, overlapping the evaluations between them.

        pipe = U8Pipe(w, h, depth=3)
        pipe.in_buffer(s)[...] = pixels           # (h, w, channels) uint8 view of pinned memory
        img = pipe.upload(s)                      # SlotImage, usable at once
        pipe.download(s, result, srgb=False)      # asynchronous
        out = pipe.wait_download(s)               # (h, w, 4) uint8 view, valid until the slot's next download"""

    def __init__(self, width, height, channels=4, depth=3):
        h = C.c_void_p()
        _check(_lib.load().kc_u8_pipe_create(int(width), int(height), int(channels), int(depth), C.byref(h)))
        self._h, self.width, self.height, self.channels, self.depth = h, int(width), int(height), int(channels), int(depth)

    def close(self):
        if self._h:
            _check(_lib.load().kc_u8_pipe_free(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _buffers(self, slot):
        a, b = C.c_void_p(), C.c_void_p()
        _check(_lib.load().kc_u8_pipe_buffers(self._h, int(slot), C.byref(a), C.byref(b)))
        return a.value, b.value

    def in_buffer(self, slot):
        p, _ = self._buffers(slot)
        n = self.width * self.height * self.channels
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(p)).reshape(self.height, self.width, self.channels)

    def out_buffer(self, slot):
        _, p = self._buffers(slot)
        n = self.width * self.height * 4
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(p)).reshape(self.height, self.width, 4)

    def upload(self, slot):
        out = C.c_void_p()
        _check(_lib.load().kc_u8_pipe_upload(self._h, int(slot), C.byref(out)))
        return SlotImage(out.value)

    def download(self, slot, image, srgb=False):
        _check(_lib.load().kc_u8_pipe_download(self._h, int(slot), image._h, int(bool(srgb))))

    def wait_download(self, slot):
        _check(_lib.load().kc_u8_pipe_wait_download(self._h, int(slot)))
        return self.out_buffer(slot)


class PartitionPolicy:
    Auto, Spread, Bands = 0, 1, 2


class PlanKind:
    Single, Branches, Bands = 0, 1, 2


class NodeKind:
    Source, Replicated, Compute = 0, 1, 2


class Partition:
    """Placement plan of one graph evaluation over `world` ranks (include/kanter_core_amd.h, "Multi-GPU").
    nodes: [(node_id, rank, component, kind)] in topological order, rank -1 = replicated;
    transfers: [(node_id, slot_id, src_rank, dst_rank, level)] in execution order."""

    def __init__(self, handle):
        from ._lib import kc_placement, kc_transfer
        L = _lib.load()
        self._h = C.c_void_p(handle)
        w, hm, lv = C.c_int(), C.c_int(), C.c_int()
        _check(L.kc_partition_info(self._h, C.byref(w), C.byref(hm), C.byref(lv)))
        self.world, self.home, self.levels = w.value, hm.value, lv.value
        n = C.c_uint32()
        _check(L.kc_partition_nodes(self._h, None, 0, C.byref(n)))
        buf = (kc_placement * max(n.value, 1))()
        _check(L.kc_partition_nodes(self._h, buf, n.value, C.byref(n)))
        self.nodes = [(buf[i].node_id, buf[i].rank, buf[i].component, buf[i].kind) for i in range(n.value)]
        _check(L.kc_partition_transfers(self._h, None, 0, C.byref(n)))
        tb = (kc_transfer * max(n.value, 1))()
        _check(L.kc_partition_transfers(self._h, tb, n.value, C.byref(n)))
        self.transfers = [(tb[i].node_id, tb[i].slot_id, tb[i].src_rank, tb[i].dst_rank, tb[i].level) for i in range(n.value)]
        k, a, b, c = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        _check(L.kc_partition_kind(self._h, C.byref(k), C.byref(a), C.byref(b), C.byref(c)))
        # kind: PlanKind; estimates: what PartitionPolicy.Auto compared, in units of one fused RGBA Mix chain over the image
        self.kind, self.estimates = k.value, {"single": a.value, "branches": b.value, "bands": c.value if c.value >= 0 else None}
        from ._lib import kc_band_range
        fw, fh = C.c_uint32(), C.c_uint32()
        _check(L.kc_partition_bands(self._h, None, 0, C.byref(n), C.byref(fw), C.byref(fh)))
        bb = (kc_band_range * max(n.value, 1))()
        _check(L.kc_partition_bands(self._h, bb, n.value, C.byref(n), C.byref(fw), C.byref(fh)))
        self.bands = [(bb[i].y0, bb[i].y1) for i in range(n.value)]  # PlanKind.Bands: rows of the requested node per rank
        self.full_size = (fw.value, fh.value)
        self.gather = True

    def set_gather(self, gather):
        """PlanKind.Bands: False leaves every rank's band where it is (evaluate_partitioned returns the band everywhere)."""
        _check(_lib.load().kc_partition_set_gather(self._h, int(bool(gather))))
        self.gather = bool(gather)

    def __del__(self):
        try:
            _lib.load().kc_partition_free(self._h)
        except Exception:
            pass

    def rank_of(self, node_id):
        return next(r for (n, r, _, _) in self.nodes if n == node_id)

    def local_nodes(self, rank):
        """Nodes this rank evaluates: its own plus the replicated ones."""
        return [n for (n, r, _, _) in self.nodes if r == rank or r == -1]


class TextureProcessor:
    """src/texture_processor.rs:18-115.  `memory_threshold` is kept for API compatibility: planes
    stay in HBM, nothing is ever spilled to disk."""

    def __init__(self, memory_threshold=10_000_000, device=None):
        if not is_initialized():
            init(device)
        h = C.c_void_p()
        _check(_lib.load().kc_tex_pro_new(int(memory_threshold), C.byref(h)))
        self._h = h
        self.memory_threshold = int(memory_threshold)

    new = classmethod(lambda cls, memory_threshold=10_000_000: cls(memory_threshold))

    def __del__(self):
        try:
            _lib.load().kc_tex_pro_free(self._h)
        except Exception:
            pass

    def new_live_graph(self):
        h = C.c_void_p()
        _check(_lib.load().kc_tex_pro_new_live_graph(self._h, C.byref(h)))
        return LiveGraph(h.value, self)

    @staticmethod
    def buffer_rgba(live_graph, node_id, slot_id):
        return live_graph.await_clean(node_id).buffer_rgba(node_id, slot_id)

    @staticmethod
    def node_slot_datas(live_graph, node_id):
        return live_graph.await_clean(node_id).node_slot_datas(node_id)

    @staticmethod
    def await_slot_data_size(live_graph, node_id, slot_id):
        return live_graph.await_clean(node_id).slot_data_size(node_id, slot_id)

    def processing_node_count(self):
        """src/texture_processor.rs:107-109: evaluation is synchronous, nothing is ever in flight
        between calls."""
        return 0

    def set_max_processing_nodes(self, count):
        """src/texture_processor.rs:111-114: admission control of the reference's thread-per-node
        scheduler; kernels are stream-ordered here, so the value is only recorded."""
        self.max_processing_nodes = int(count)
