"""Row-band planning on the host (no GPU): which rows of every source image a band of the result reads.  Checked against
the closed forms: pointwise nodes pass the band through, HeightToNormal adds the row above (toroidal), an implicit resize
needs [left(y0), left(y1 - 1) + count(y1 - 1)) of its source, with the tap windows taken from the oracle's restatement of
the resampler (tests may use the oracle)."""
import json

import pytest

import kanter_core_amd as kc
from golden_graphs import G
from test_multi_gpu_gloo import HostOnlyTexPro


def graph_resize_h2n(filt="Triangle"):
    """big (Embed 0, 64 x 48 rows) + small (Embed 1, 16 x 12 rows, resized by the Mix) -> Separate.R -> HeightToNormal"""
    g = G()
    big, small = g.add({"Embed": 0}), g.add({"Embed": 1})
    mix = g.add({"Mix": "Add"}, filt=filt)
    g.connect(big, mix, 0, 0)
    g.connect(small, mix, 0, 1)
    sep = g.add("SeparateRgba")
    g.connect(mix, sep, 0, 0)
    h2n = g.add("HeightToNormal")
    g.connect(sep, h2n, 0, 0)
    out = g.add({"OutputRgba": "out"})
    g.connect(h2n, out, 0, 0)
    return g.dict(), dict(big=big, small=small, mix=mix, h2n=h2n, out=out)


class FakeImage:
    """A host-only stand-in for embedded images: band planning only looks at sizes."""


def live_graph_with_sizes(graph, sizes):
    import ctypes as C
    from kanter_core_amd import _lib
    L = _lib.load()
    lg = HostOnlyTexPro().new_live_graph()
    lg.set_node_graph(kc.NodeGraph.from_json(json.dumps(graph)))
    keep = []
    for eid, (w, h) in sizes.items():
        planes = []
        for _ in range(4):
            p = C.c_void_p()
            assert L.kc_plane_const(w, h, 0.5, C.byref(p)) == 0  # constant planes need no device
            planes.append(p)
        im = C.c_void_p()
        assert L.kc_image_rgba((C.c_void_p * 4)(*[p.value for p in planes]), C.byref(im)) == 0
        for p in planes:
            L.kc_plane_release(p)
        img = kc.SlotImage(im.value)
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, img), eid)
        keep.append(img)
    return lg, keep


@pytest.mark.parametrize("filt", ["Triangle", "Lanczos3", "Nearest"])
def test_band_source_rows_follow_the_node_types(filt):
    from oracle import oracle as orc
    graph, nm = graph_resize_h2n(filt)
    lg, keep = live_graph_with_sizes(graph, {0: (64, 48), 1: (16, 12)})
    left, count, _ = orc.resize_taps(12, 48, filt)
    for (y0, y1) in ((10, 20), (1, 2), (24, 48), (47, 48)):
        rows = lg.band_source_rows(nm["out"], y0, y1)
        a = y0 - 1  # HeightToNormal reads the row above
        assert rows[nm["big"]] == (a, y1, 64, 48)
        lo = min(int(left[y]) for y in range(a, y1))
        hi = max(int(left[y]) + int(count[y]) for y in range(a, y1))
        assert rows[nm["small"]] == (lo, hi, 16, 12)
    # the band that starts at row 0 needs the image's LAST row (row -1) of the pointwise input; a resized input
    # whose output band wraps is needed whole (it is the small one)
    rows = lg.band_source_rows(nm["out"], 0, 16)
    assert rows[nm["big"]] == (-1, 16, 64, 48) and rows[nm["small"]] == (0, 12, 16, 12)
    # the whole image is the whole image
    rows = lg.band_source_rows(nm["out"], 0, 48)
    assert rows[nm["big"]] == (0, 48, 64, 48) and rows[nm["small"]] == (0, 12, 16, 12)


def test_band_arguments_are_checked():
    graph, nm = graph_resize_h2n()
    lg, keep = live_graph_with_sizes(graph, {0: (64, 48), 1: (16, 12)})
    for (y0, y1) in ((-1, 4), (4, 4), (5, 3), (0, 49)):
        with pytest.raises(kc.TexProError):
            lg.band_source_rows(nm["out"], y0, y1)
    with pytest.raises(kc.TexProError):
        lg.band_source_rows(987654, 0, 4)
    # a Graph node on the path is refused rather than mis-evaluated
    g = G()
    inner = G()
    inner.add({"Value": 1.0})
    gn = g.add({"Graph": inner.dict()})
    lg2 = HostOnlyTexPro().new_live_graph()
    lg2.set_node_graph(kc.NodeGraph.from_json(json.dumps(g.dict())))
    with pytest.raises(kc.TexProError):
        lg2.band_source_rows(gn, 0, 1)
