"""ctypes front end of the CPU oracle (oracle/kc_oracle.c) plus a literal restatement of the
reference's node dispatch (`process_node`) used to evaluate whole graphs on the CPU.

TEST INFRASTRUCTURE ONLY -- see the header of kc_oracle.c.  Nothing under kanter_core_amd/
imports this module; tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

Reference citations are relative to /root/reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkc_oracle.so")

MIX_TYPES = {"Add": 0, "Subtract": 1, "Multiply": 2, "Divide": 3, "Pow": 4}
FILTERS = {"Nearest": 0, "Triangle": 1, "CatmullRom": 2, "Gaussian": 3, "Lanczos3": 4}
POLICIES = {"MostPixels": 0, "LeastPixels": 1, "LargestAxes": 2, "SmallestAxes": 3, "SpecificSlot": 4,
            "SpecificSize": 5}


def build(force=False):
    """Compile oracle/libkc_oracle.so with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "kc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkc_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        u8p = C.POINTER(C.c_uint8)
        u32p = C.POINTER(C.c_uint32)
        L.kco_set_threads.argtypes = [C.c_int]
        L.kco_set_threads.restype = C.c_int
        L.kco_max_threads.restype = C.c_int
        L.kco_mix_plane.argtypes = [C.c_int, fp, fp, fp, C.c_size_t]
        L.kco_mix_plane.restype = C.c_int
        L.kco_fill.argtypes = [fp, C.c_size_t, C.c_float]
        L.kco_fill.restype = None
        L.kco_rgba_to_gray.argtypes = [fp, fp, fp, fp, C.c_size_t]
        L.kco_rgba_to_gray.restype = None
        L.kco_resize_plane.argtypes = [fp, C.c_uint32, C.c_uint32, fp, C.c_uint32, C.c_uint32, C.c_int]
        L.kco_resize_plane.restype = C.c_int
        L.kco_resize_taps.argtypes = [C.c_uint32, C.c_uint32, C.c_int, u32p, u32p, fp, C.c_uint32]
        L.kco_resize_taps.restype = C.c_int
        L.kco_calculate_size.argtypes = [C.c_int, u32p, u32p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, u32p, u32p]
        L.kco_calculate_size.restype = C.c_int
        L.kco_height_to_normal.argtypes = [fp, C.c_uint32, C.c_uint32, fp, fp, fp]
        L.kco_height_to_normal.restype = None
        L.kco_deconstruct_u8.argtypes = [u8p, C.c_size_t, C.c_int, fp, fp, fp, fp]
        L.kco_deconstruct_u8.restype = None
        L.kco_to_u8_rgba.argtypes = [fp, fp, fp, fp, C.c_size_t, C.c_int, u8p]
        L.kco_to_u8_rgba.restype = None
        L.kco_to_u8_gray.argtypes = [fp, C.c_size_t, C.c_int, u8p]
        L.kco_to_u8_gray.restype = None
        L.kco_chain32.argtypes = [C.POINTER(fp), C.POINTER(fp), C.POINTER(fp), C.c_uint32, C.c_uint32, C.c_int]
        L.kco_chain32.restype = C.c_int
        L.kco_set_plane_pool.argtypes = [C.c_int]
        L.kco_set_plane_pool.restype = C.c_int
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _plane(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2
    return a


def set_threads(n):
    return lib().kco_set_threads(int(n))


def set_plane_pool(on):
    """chain32 only: planes released by a node are reused by the next (1) instead of freed and allocated afresh as the
    reference does (0, the default).  Same arithmetic; what lets the many-core run scale."""
    return lib().kco_set_plane_pool(int(bool(on)))


def max_threads():
    return lib().kco_max_threads()


# ------------------------------------------------------------------ plane-level ops
def mix_plane(op, left, right):
    """src/node/mix.rs:136-192"""
    left, right = _plane(left), _plane(right)
    assert left.shape == right.shape
    out = np.empty_like(left)
    op = MIX_TYPES[op] if isinstance(op, str) else int(op)
    rc = lib().kco_mix_plane(op, _fp(left), _fp(right), _fp(out), left.size)
    assert rc == 0
    return out


def fill(h, w, v):
    """vec![v; n] -- src/slot_image.rs:28-64"""
    out = np.empty((h, w), np.float32)
    lib().kco_fill(_fp(out), out.size, float(v))
    return out


def rgba_to_gray(r, g, b):
    """src/slot_image.rs:242-253"""
    r, g, b = _plane(r), _plane(g), _plane(b)
    out = np.empty_like(r)
    lib().kco_rgba_to_gray(_fp(r), _fp(g), _fp(b), _fp(out), r.size)
    return out


def resize_plane(src, width, height, filt="Triangle"):
    """image::imageops::resize as called from src/shared.rs:159-199"""
    src = _plane(src)
    out = np.empty((height, width), np.float32)
    f = FILTERS[filt] if isinstance(filt, str) else int(filt)
    rc = lib().kco_resize_plane(_fp(src), src.shape[1], src.shape[0], _fp(out), width, height, f)
    assert rc == 0
    return out


def resize_taps(in_n, out_n, filt):
    """Tap table (left, count, normalised weights) of one sampling axis."""
    f = FILTERS[filt] if isinstance(filt, str) else int(filt)
    stride = lib().kco_resize_taps(in_n, out_n, f, None, None, None, 0)
    assert stride > 0
    left = np.zeros(out_n, np.uint32)
    count = np.zeros(out_n, np.uint32)
    w = np.zeros((out_n, stride), np.float32)
    u32p = C.POINTER(C.c_uint32)
    rc = lib().kco_resize_taps(in_n, out_n, f, left.ctypes.data_as(u32p), count.ctypes.data_as(u32p), _fp(w), stride)
    assert rc == stride
    return left, count, w


def calculate_size(policy, sizes, slot_index=-1, spec=(0, 0)):
    """src/shared.rs:61-139; sizes = [(w, h)] in edge insertion order."""
    p = POLICIES[policy] if isinstance(policy, str) else int(policy)
    n = len(sizes)
    ws = (C.c_uint32 * max(n, 1))(*[s[0] for s in sizes])
    hs = (C.c_uint32 * max(n, 1))(*[s[1] for s in sizes])
    ow, oh = C.c_uint32(), C.c_uint32()
    rc = lib().kco_calculate_size(p, ws, hs, n, slot_index, spec[0], spec[1], C.byref(ow), C.byref(oh))
    if rc != 0:
        raise ValueError("calculate_size failed (reference would panic)")
    return ow.value, oh.value


def height_to_normal(hgt):
    """src/node/height_to_normal.rs:16-77"""
    hgt = _plane(hgt)
    nx, ny, nz = np.empty_like(hgt), np.empty_like(hgt), np.empty_like(hgt)
    lib().kco_height_to_normal(_fp(hgt), hgt.shape[1], hgt.shape[0], _fp(nx), _fp(ny), _fp(nz))
    return nx, ny, nz


def deconstruct_u8(px):
    """src/shared.rs:16-56; px = uint8 array (h, w, channels)."""
    px = np.ascontiguousarray(px, dtype=np.uint8)
    if px.ndim == 2:
        px = px[:, :, None]
    h, w, c = px.shape
    planes = [np.empty((h, w), np.float32) for _ in range(4)]
    lib().kco_deconstruct_u8(px.ctypes.data_as(C.POINTER(C.c_uint8)), h * w, c, *[_fp(p) for p in planes])
    return planes


def to_u8(image, srgb=False):
    """SlotImage::to_u8 / to_u8_srgb, src/slot_image.rs:141-207 -> uint8 (h, w, 4)."""
    planes = [_plane(p) for p in image.planes]
    h, w = planes[0].shape
    out = np.empty((h, w, 4), np.uint8)
    o = out.ctypes.data_as(C.POINTER(C.c_uint8))
    if image.is_rgba:
        lib().kco_to_u8_rgba(*[_fp(p) for p in planes], h * w, int(srgb), o)
    else:
        lib().kco_to_u8_gray(_fp(planes[0]), h * w, int(srgb), o)
    return out


def chain32(a, b, n_nodes=32):
    """CPU baseline: the 32-node linear mix/invert graph evaluated node by node (kco_chain32)."""
    a = [_plane(p) for p in a[:3]]
    b = [_plane(p) for p in b[:3]]
    h, w = a[0].shape
    out = [np.empty((h, w), np.float32) for _ in range(4)]
    fp = C.POINTER(C.c_float)
    aa = (fp * 3)(*[_fp(p) for p in a])
    bb = (fp * 3)(*[_fp(p) for p in b])
    oo = (fp * 4)(*[_fp(p) for p in out])
    rc = lib().kco_chain32(aa, bb, oo, w, h, n_nodes)
    assert rc == 0
    return out


# ------------------------------------------------------------------ SlotImage / SlotData model
class Image:
    """SlotImage (src/slot_image.rs:15-19): Gray = 1 plane, Rgba = 4 planes; planes are shared
    by reference exactly where the reference clones an Arc."""

    def __init__(self, planes):
        assert len(planes) in (1, 4)
        self.planes = list(planes)

    @property
    def is_rgba(self):
        return len(self.planes) == 4

    @property
    def size(self):
        h, w = self.planes[0].shape
        return (w, h)

    @staticmethod
    def from_value(size, value, rgba):
        """src/slot_image.rs:28-64"""
        w, h = size
        if rgba:
            return Image([fill(h, w, value), fill(h, w, value), fill(h, w, value), fill(h, w, 1.0)])
        return Image([fill(h, w, value)])

    def as_type(self, rgba):
        """src/slot_image.rs:212-256"""
        if self.is_rgba == rgba:
            return self
        w, h = self.size
        if not self.is_rgba:
            p = self.planes[0]
            return Image([p, p, p, fill(h, w, 1.0)])
        return Image([rgba_to_gray(*self.planes[:3])])


def pixel_buffer(v):
    """src/node/mod.rs:240-244"""
    return np.full((1, 1), v, np.float32)


class SlotData:
    """src/slot_data.rs:34-39"""

    def __init__(self, node_id, slot_id, image):
        self.node_id, self.slot_id, self.image = node_id, slot_id, image

    @property
    def size(self):
        return self.image.size


def image_from_u8(px):
    """read_slot_image, src/shared.rs:218-261: always RGBA (missing channels defaulted)."""
    return Image(deconstruct_u8(px))


# ------------------------------------------------------------------ node table (src/node/node_type.rs:140-211)
def _node_kind(node):
    nt = node["node_type"]
    if isinstance(nt, str):
        return nt, None
    (k, v), = nt.items()
    return k, v


def input_slots(node):
    kind, arg = _node_kind(node)
    if kind in ("OutputGray", "OutputRgba"):
        return [("input", 0)]
    if kind == "Graph":
        return [(_node_kind(n)[1], n["node_id"]) for n in arg["nodes"] if _node_kind(n)[0] in ("InputGray", "InputRgba")]
    if kind == "Mix":
        return [("left", 0), ("right", 1)]
    if kind in ("HeightToNormal", "SeparateRgba"):
        return [("input", 0)]
    if kind == "CombineRgba":
        return [("red", 0), ("green", 1), ("blue", 2), ("alpha", 3)]
    return []


def output_slot_count(node):
    kind, arg = _node_kind(node)
    if kind in ("OutputGray", "OutputRgba"):
        return 0
    if kind == "Graph":
        return len([n for n in arg["nodes"] if _node_kind(n)[0] in ("OutputGray", "OutputRgba")])
    if kind == "SeparateRgba":
        return 4
    return 1


def _with_slot(slot_datas, slot_id):
    """slot_data_with_slot_id, src/node/process_shared.rs:26-36"""
    for sd in slot_datas:
        if sd.slot_id == slot_id:
            return sd
    return None


# ------------------------------------------------------------------ resize pre-step (src/shared.rs:61-216)
def _policy(node):
    p = node.get("resize_policy", "MostPixels")
    if isinstance(p, str):
        return p, None
    (k, v), = p.items()
    return k, v


def ref_calculate_size(slot_datas, edges_sorted, node):
    kind, arg = _policy(node)
    sizes = [sd.size for sd in slot_datas]
    if kind == "SpecificSize":
        return (arg["width"], arg["height"])
    if kind == "SpecificSlot":
        edge = next((e for e in edges_sorted if e["input_slot"] == arg), None)
        if edge is None and edges_sorted:
            edge = edges_sorted[0]
        idx = -1
        if edge is not None:
            idx = next(i for i, sd in enumerate(slot_datas)
                       if sd.slot_id == edge["output_slot"] and sd.node_id == edge["output_id"])
        return calculate_size("SpecificSlot", sizes, slot_index=idx)
    return calculate_size(kind, sizes)


def ref_resize_buffers(slot_datas, edges_sorted, node):
    if not slot_datas:
        return list(slot_datas)
    size = ref_calculate_size(slot_datas, edges_sorted, node)
    filt = node.get("resize_filter", "Triangle")
    out = []
    for sd in slot_datas:
        if sd.size != size:
            planes = [resize_plane(p, size[0], size[1], filt) for p in sd.image.planes]
            out.append(SlotData(sd.node_id, sd.slot_id, Image(planes)))
        else:
            out.append(sd)
    return out


# ------------------------------------------------------------------ per-node process fns
def mix_process(slot_datas, node_id, mix_type):
    """src/node/mix.rs:51-134"""
    left_sd, right_sd = _with_slot(slot_datas, 0), _with_slot(slot_datas, 1)
    if left_sd is not None:
        rgba = left_sd.image.is_rgba
        right = right_sd.image.as_type(rgba) if right_sd is not None else Image.from_value(left_sd.size, 0.0, rgba)
        left = left_sd.image
    elif right_sd is not None:
        left = Image.from_value(right_sd.size, 0.0, right_sd.image.is_rgba)
        right = right_sd.image
    else:
        return [SlotData(node_id, 0, Image.from_value((1, 1), 0.0, False))]
    w, h = left.size
    if left.is_rgba != right.is_rgba:
        return []
    if left.is_rgba:
        planes = [mix_plane(mix_type, left.planes[c], right.planes[c]) for c in range(3)] + [fill(h, w, 1.0)]
    else:
        planes = [mix_plane(mix_type, left.planes[0], right.planes[0])]
    return [SlotData(node_id, 0, Image(planes))]


def separate_process(slot_datas, node_id):
    """src/node/separate_rgba.rs:38-69"""
    if slot_datas and slot_datas[0].image.is_rgba:
        return [SlotData(node_id, i, Image([slot_datas[0].image.planes[i]])) for i in range(4)]
    return [SlotData(node_id, i, Image([pixel_buffer(0.0)])) for i in range(4)]


def combine_process(slot_datas, node_id):
    """src/node/combine_rgba.rs:14-97"""
    size = slot_datas[0].size if slot_datas else (1, 1)
    shared_zero = [None]

    def default(alpha):
        if not alpha and shared_zero[0] is not None:
            return shared_zero[0]
        p = fill(size[1], size[0], 1.0 if alpha else 0.0)
        shared_zero[0] = p  # combine_rgba.rs:58: the alpha default also overwrites `existing_buffer`
        return p

    planes = []
    for slot in range(4):
        d = default(slot == 3)
        sd = _with_slot(slot_datas, slot)
        if sd is not None:
            assert not sd.image.is_rgba, "It shouldn't be possible to connect an RGBA image into this slot"
            planes.append(sd.image.planes[0])
        else:
            planes.append(d)
    return [SlotData(node_id, 0, Image(planes))]


def h2n_process(slot_datas, node_id):
    """src/node/height_to_normal.rs:16-77"""
    sd = _with_slot(slot_datas, 0)
    if sd is None or sd.image.is_rgba:
        return []
    nx, ny, nz = height_to_normal(sd.image.planes[0])
    h, w = nx.shape
    return [SlotData(node_id, 0, Image([nx, ny, nz, fill(h, w, 1.0)]))]


def output_process(slot_datas, node_id, kind):
    """src/node/output.rs:12-33"""
    if slot_datas:
        return [SlotData(node_id, 0, slot_datas[0].image)]
    if kind == "OutputRgba":
        return [SlotData(node_id, 0, Image([pixel_buffer(0.0), pixel_buffer(0.0), pixel_buffer(0.0), pixel_buffer(1.0)]))]
    return [SlotData(node_id, 0, Image([pixel_buffer(0.0)]))]


class RefGraph:
    """Evaluates a NodeGraph (the reference's serde JSON shape: {"nodes": [...], "edges": [...]},
    src/node_graph.rs:16-22) the way engine::process_loop + process_node do, one node at a time,
    on the CPU oracle.  `load_image(path) -> uint8 (h, w, c)` resolves Image nodes;
    `input_slot_datas` feeds Input* nodes (src/node/input_*.rs); `embedded` maps
    EmbeddedSlotDataId -> Image (src/node/embed.rs:33-50)."""

    def __init__(self, graph, load_image=None, input_slot_datas=(), embedded=None):
        self.nodes = {n["node_id"]: n for n in graph["nodes"]}
        self.edges = list(graph["edges"])
        self.load_image = load_image
        self.input_slot_datas = list(input_slot_datas)
        self.embedded = dict(embedded or {})
        self.results = {}

    def node_slot_datas(self, node_id):
        if node_id not in self.results:
            self.results[node_id] = self._process(node_id)
        return self.results[node_id]

    def slot_data(self, node_id, slot_id):
        sd = _with_slot(self.node_slot_datas(node_id), slot_id)
        if sd is None:
            raise KeyError("NoSlotData")
        return sd

    def buffer_rgba(self, node_id, slot_id=0):
        return to_u8(self.slot_data(node_id, slot_id).image)

    def _process(self, node_id):
        node = self.nodes[node_id]
        # engine.rs:213-218: input edges in insertion order; :261-275 one SlotData per edge
        edges = [e for e in self.edges if e["input_id"] == node_id]
        inputs = [self.slot_data(e["output_id"], e["output_slot"]) for e in edges]
        # node_type.rs:229-237
        edges_sorted = sorted(edges, key=lambda e: e["input_slot"])
        resized = ref_resize_buffers(inputs, edges_sorted, node)
        # assign_slot_ids, node_type.rs:250-267
        slot_datas = []
        for e in edges_sorted:
            sd = next(s for s in resized if s.slot_id == e["output_slot"] and s.node_id == e["output_id"])
            slot_datas.append(SlotData(e["input_id"], e["input_slot"], sd.image))
        out = self._dispatch(node, slot_datas)
        kind, _ = _node_kind(node)
        if kind not in ("OutputGray", "OutputRgba") and len(out) != output_slot_count(node):
            raise RuntimeError("InvalidBufferCount")  # node_type.rs:124-137
        return out

    def _dispatch(self, node, slot_datas):
        """process_node_internal, src/node/node_type.rs:107-122"""
        kind, arg = _node_kind(node)
        nid = node["node_id"]
        if kind in ("InputGray", "InputRgba"):
            sd = next((s for s in self.input_slot_datas if s.node_id == nid), None)
            if kind == "InputRgba":
                return [SlotData(nid, 0, self.input_slot_datas[0].image)]  # input_rgba.rs:7-13
            return [sd] if sd is not None else []  # input_gray.rs:7-16
        if kind in ("OutputGray", "OutputRgba"):
            return output_process(slot_datas, nid, kind)
        if kind == "Graph":
            # src/node/graph.rs:14-51
            child = RefGraph(arg, self.load_image,
                             [SlotData(sd.slot_id, 0, sd.image) for sd in slot_datas], self.embedded)
            out = []
            for n in arg["nodes"]:
                if _node_kind(n)[0] in ("OutputGray", "OutputRgba"):
                    for sd in child.node_slot_datas(n["node_id"]):
                        out.append(SlotData(nid, n["node_id"], sd.image))
            return out
        if kind == "Image":
            try:
                img = image_from_u8(self.load_image(arg))
            except Exception:
                img = Image([pixel_buffer(1.0), pixel_buffer(0.0), pixel_buffer(1.0), pixel_buffer(1.0)])
            return [SlotData(nid, 0, img)]
        if kind == "Embed":
            if arg not in self.embedded:
                raise RuntimeError("NodeProcessing")
            return [SlotData(nid, 0, self.embedded[arg])]
        if kind == "Value":
            return [SlotData(nid, 0, Image([pixel_buffer(arg)]))]  # value.rs:14-26
        if kind == "Mix":
            return mix_process(slot_datas, nid, arg)
        if kind == "HeightToNormal":
            return h2n_process(slot_datas, nid)
        if kind == "SeparateRgba":
            return separate_process(slot_datas, nid)
        if kind == "CombineRgba":
            return combine_process(slot_datas, nid)
        raise NotImplementedError(kind)
