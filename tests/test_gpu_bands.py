"""Row-band evaluation on the device: the bands of a result, evaluated one after the other on one GPU (what N ranks do
concurrently), stacked, must equal the whole-image evaluation BIT FOR BIT -- for pointwise graphs, graphs with implicit
resizes (up- and down-sampling, five filters), HeightToNormal (1-row toroidal halo: the first band needs the last row),
and sources that are themselves sharded by rows and embedded with exactly the halo the planner asks for."""
import json

import numpy as np
import pytest

from golden_graphs import G
from util import SEED_A, SEED_B, assert_planes, splitmix_plane

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    return kc


def build(kc, graph, images, bands=None):
    """LiveGraph of `graph`; images: {embed id: [planes]}; bands: {embed id: (y0, y1)} embeds only those rows."""
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    lg.set_node_graph(kc.NodeGraph.from_json(json.dumps(graph)))
    for eid, planes in images.items():
        if bands is None or eid not in bands:
            lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(planes)), eid)
        else:
            y0, y1 = bands[eid]
            h = planes[0].shape[0]
            rows = [r % h for r in range(y0, y1)]  # y0 < 0: the wrapped rows come first
            lg.embed_slot_data_band(kc.SlotData(0, 0, kc.SlotImage.from_planes([p[rows] for p in planes])), eid, y0, h)
    return tp, lg


def splits(h):
    return [[(0, h)], [(0, h // 2), (h // 2, h)], [(0, 1), (1, h // 3), (h // 3, h - 1), (h - 1, h)],
            [(y, min(y + 7, h)) for y in range(0, h, 7)]]


def check_bands(kc, graph, images, root, what):
    tp, lg = build(kc, graph, images)
    whole = lg.await_clean(root).slot_data(root, 0).image.planes()
    h = whole[0].shape[0]
    for split in splits(h):
        tp2, lg2 = build(kc, graph, images)
        parts = [lg2.evaluate_band(root, y0, y1).planes() for (y0, y1) in split]
        assert all(p[0].shape[0] == y1 - y0 for p, (y0, y1) in zip(parts, split))
        stacked = [np.concatenate([p[c] for p in parts], axis=0) for c in range(len(whole))]
        assert_planes(stacked, whole, what="%s, %d bands" % (what, len(split)))
    return whole


def chain_graph(n_nodes=12):
    g = G()
    a, b = g.add({"Embed": 0}), g.add({"Embed": 1})
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
    return g.dict(), prev


def test_pointwise_chain_bands(kc):
    h, w = 61, 200
    imgs = {0: [splitmix_plane(SEED_A, c, h, w) for c in range(4)], 1: [splitmix_plane(SEED_B, c, h, w) for c in range(4)]}
    graph, root = chain_graph()
    check_bands(kc, graph, imgs, root, "chain")


@pytest.mark.parametrize("filt", ["Nearest", "Triangle", "CatmullRom", "Gaussian", "Lanczos3"])
@pytest.mark.parametrize("sizes", [((48, 64), (12, 16)), ((40, 72), (131, 200)), ((57, 33), (57, 90))])
def test_resize_and_height_to_normal_bands(kc, filt, sizes):
    """big + small -> Mix (the smaller / larger one is resampled to the other's size, both axes or one) -> Separate ->
    HeightToNormal(R) -> Mix(Multiply) with the blend -> Output: resize halos and the toroidal 1-row halo together."""
    (h0, w0), (h1, w1) = sizes
    g = G()
    e0, e1 = g.add({"Embed": 0}), g.add({"Embed": 1})
    mix = g.add({"Mix": "Add"}, filt=filt)
    g.connect(e0, mix, 0, 0)
    g.connect(e1, mix, 0, 1)
    sep = g.add("SeparateRgba")
    g.connect(mix, sep, 0, 0)
    h2n = g.add("HeightToNormal")
    g.connect(sep, h2n, 1, 0)
    fin = g.add({"Mix": "Multiply"}, filt=filt)
    g.connect(h2n, fin, 0, 0)
    g.connect(mix, fin, 0, 1)
    out = g.add({"OutputRgba": "out"})
    g.connect(fin, out, 0, 0)
    imgs = {0: [splitmix_plane(SEED_A, c, h0, w0) for c in range(4)], 1: [splitmix_plane(SEED_B, c, h1, w1) for c in range(4)]}
    check_bands(kc, g.dict(), imgs, out, "%s %s" % (filt, sizes))


def test_two_height_to_normal_nodes_in_a_row(kc):
    """Halo of 2: the second node's band needs one more row of the first node's output, which needs one more of its input."""
    h, w = 37, 52
    g = G()
    e0 = g.add({"Embed": 0})
    s1 = g.add("SeparateRgba")
    g.connect(e0, s1, 0, 0)
    n1 = g.add("HeightToNormal")
    g.connect(s1, n1, 0, 0)
    s2 = g.add("SeparateRgba")
    g.connect(n1, s2, 0, 0)
    n2 = g.add("HeightToNormal")
    g.connect(s2, n2, 2, 0)
    imgs = {0: [splitmix_plane(SEED_A, c, h, w) for c in range(4)]}
    check_bands(kc, g.dict(), imgs, n2, "h2n x2")


def test_sharded_sources_with_exactly_the_planned_halo(kc):
    """Each 'rank' embeds only the rows kc_live_graph_band_source_rows asks for -- including the wrapped last row for the
    band that starts at row 0 -- and still reproduces its band of the whole-image result."""
    (h0, w0), (h1, w1) = (64, 96), (16, 24)
    g = G()
    e0, e1 = g.add({"Embed": 0}), g.add({"Embed": 1})
    mix = g.add({"Mix": "Subtract"}, filt="CatmullRom")
    g.connect(e0, mix, 0, 0)
    g.connect(e1, mix, 0, 1)
    sep = g.add("SeparateRgba")
    g.connect(mix, sep, 0, 0)
    h2n = g.add("HeightToNormal")
    g.connect(sep, h2n, 0, 0)
    graph = g.dict()
    imgs = {0: [splitmix_plane(SEED_A, c, h0, w0) for c in range(4)], 1: [splitmix_plane(SEED_B, c, h1, w1) for c in range(4)]}
    tp, lg = build(kc, graph, imgs)
    whole = lg.await_clean(h2n).slot_data(h2n, 0).image.planes()
    embed_of = {e0: 0, e1: 1}
    for (y0, y1) in ((0, 16), (16, 40), (40, 64)):
        need = lg.band_source_rows(h2n, y0, y1)
        bands = {embed_of[n]: (a, b) for n, (a, b, _, _) in need.items()}
        if y0 == 0:
            assert bands[0] == (-1, 16)
        else:
            assert bands[0] == (y0 - 1, y1) and bands[1][1] - bands[1][0] < h1
        tp2, lg2 = build(kc, graph, imgs, bands=bands)
        got = lg2.evaluate_band(h2n, y0, y1).planes()
        assert_planes(got, [p[y0:y1] for p in whole], what="sharded sources %d:%d" % (y0, y1))
        # one row less than planned is refused, not silently wrong
        if y0 > 0:
            short = dict(bands)
            short[0] = (bands[0][0] + 1, bands[0][1])
            tp3, lg3 = build(kc, graph, imgs, bands=short)
            with pytest.raises(kc.TexProError):
                lg3.evaluate_band(h2n, y0, y1)


def test_band_at_full_size_4096_with_resize_and_h2n(kc):
    """BASELINE size: 512^2 -> 4096^2 Triangle resize feeding a blend and a HeightToNormal, two bands vs the whole image."""
    S, s = 4096, 512
    g = G()
    e0, e1 = g.add({"Embed": 0}), g.add({"Embed": 1})
    mix = g.add({"Mix": "Add"})
    g.connect(e0, mix, 0, 0)
    g.connect(e1, mix, 0, 1)
    sep = g.add("SeparateRgba")
    g.connect(mix, sep, 0, 0)
    h2n = g.add("HeightToNormal")
    g.connect(sep, h2n, 0, 0)
    imgs = {0: [splitmix_plane(SEED_A, c, S, S) for c in range(4)], 1: [splitmix_plane(SEED_B, c, s, s) for c in range(4)]}
    tp, lg = build(kc, g.dict(), imgs)
    whole = lg.await_clean(h2n).slot_data(h2n, 0).image.planes()
    parts = [lg.evaluate_band(h2n, y0, y1).planes() for (y0, y1) in ((0, 2048), (2048, 4096))]
    assert_planes([np.concatenate([p[c] for p in parts], axis=0) for c in range(4)], whole, what="4096 bands")


def test_separate_of_a_gray_image_is_1x1_in_bands_too(kc):
    """SeparateRgba of a gray image returns four 1 x 1 zeros whatever the input's size (separate_rgba.rs:38-69); the band
    planner used to give them the input's size -- a wrong plan for everything downstream (found by profiles/soak_fuzz.py).
    Downstream here: a Mix of the separated channel (1 x 1) with a full-size image, resized by the Mix, then HeightToNormal."""
    h, w = 23, 40
    imgs = {0: [splitmix_plane(SEED_A, 0, h, w)], 1: [splitmix_plane(SEED_B, 0, h, w)]}
    g = G()
    a, b = g.add({"Embed": 0}), g.add({"Embed": 1})
    sep = g.add("SeparateRgba")
    g.connect(a, sep, 0, 0)
    mix = g.add({"Mix": "Add"}, filt="Lanczos3")
    g.connect(sep, mix, 2, 0)
    g.connect(b, mix, 0, 1)
    h2n = g.add("HeightToNormal")
    g.connect(mix, h2n, 0, 0)
    _, lg = build(kc, g.dict(), imgs)
    whole = lg.await_clean(sep).slot_data(sep, 2).image.planes()
    _, lg2 = build(kc, g.dict(), imgs)
    band = lg2.evaluate_band(sep, 0, 1, 2).planes()
    assert [p.shape for p in band] == [p.shape for p in whole] == [(1, 1)]
    assert_planes(band, whole, what="separate of gray")
    check_bands(kc, g.dict(), imgs, h2n, "separate of gray -> mix -> h2n")


# ---- Graph nodes (src/node/graph.rs:14-51): expanded before the band walk (csrc/bands.cpp, expand_graph_nodes) ----
def _inner_invert():
    g = G()
    white = g.add({"Value": 1.0})
    inp = g.add({"InputGray": "in"})
    sub = g.add({"Mix": "Subtract"})
    out = g.add({"OutputGray": "out"})
    g.connect(white, sub, 0, 0)
    g.connect(inp, sub, 0, 1)
    g.connect(sub, out, 0, 0)
    return g.dict(), inp, out


def test_bands_through_a_graph_node_config0_shape(kc):
    """BASELINE config #0's shape -- Image -> SeparateRgba -> Graph(invert) -> OutputGray -- with the image embedded:
    bands == whole image == oracle == 1 - R."""
    from oracle import oracle as orc
    inner, inp, outn = _inner_invert()
    g = G()
    src = g.add({"Embed": 0})
    sep = g.add("SeparateRgba")
    gn = g.add({"Graph": inner})
    out = g.add({"OutputGray": "out"})
    g.connect(src, sep, 0, 0)
    g.connect(sep, gn, 0, inp)
    g.connect(gn, out, outn, 0)
    h, w = 37, 52
    planes = [splitmix_plane(SEED_A, c, h, w) for c in range(4)]
    whole = check_bands(kc, g.dict(), {0: planes}, out, "config #0 shape")
    want = orc.RefGraph(g.dict(), embedded={0: orc.Image(planes)}).slot_data(out, 0).image.planes
    assert_planes(whole, want, what="config #0 shape vs oracle")
    assert_planes(whole, [np.float32(1.0) - planes[0]], what="closed form")
    # the band of the Graph node itself, by its output slot
    tp, lg = build(kc, g.dict(), {0: planes})
    part = lg.evaluate_band(gn, 5, 20, slot_id=outn).planes()
    assert_planes(part, [(np.float32(1.0) - planes[0])[5:20]], what="band of the Graph node's own slot")
    rows = lg.band_source_rows(out, 5, 20)
    assert rows == {src: (5, 20, w, h)}


def test_bands_through_a_graph_node_that_resizes_its_inputs(kc):
    """The Graph node's own resize (src/node/node_type.rs:229-237) happens before its graph sees the inputs: a 16 x 12 and a
    64 x 48 input under MostPixels / CatmullRom, two Input nodes, a HeightToNormal inside (halo through the expansion)."""
    from oracle import oracle as orc
    ig = G()
    ia = ig.add({"InputGray": "a"})
    ib = ig.add({"InputGray": "b"})
    mul = ig.add({"Mix": "Multiply"})
    h2n = ig.add("HeightToNormal")
    o1 = ig.add({"OutputRgba": "normal"})
    o2 = ig.add({"OutputGray": "product"})
    ig.connect(ia, mul, 0, 0)
    ig.connect(ib, mul, 0, 1)
    ig.connect(mul, h2n, 0, 0)
    ig.connect(h2n, o1, 0, 0)
    ig.connect(mul, o2, 0, 0)
    g = G()
    small, big = g.add({"Embed": 0}), g.add({"Embed": 1})
    s1, s2 = g.add("SeparateRgba"), g.add("SeparateRgba")
    gn = g.add({"Graph": ig.dict()}, filt="CatmullRom")
    mix = g.add({"Mix": "Add"})
    out = g.add({"OutputRgba": "out"})
    g.connect(small, s1, 0, 0)
    g.connect(big, s2, 0, 0)
    g.connect(s1, gn, 1, ia)
    g.connect(s2, gn, 2, ib)
    g.connect(gn, mix, o1, 0)
    g.connect(big, mix, 0, 1)
    g.connect(mix, out, 0, 0)
    a = [splitmix_plane(SEED_A, c, 12, 16) for c in range(4)]
    b = [splitmix_plane(SEED_B, c, 48, 64) for c in range(4)]
    whole = check_bands(kc, g.dict(), {0: a, 1: b}, out, "resizing Graph node")
    want = orc.RefGraph(g.dict(), embedded={0: orc.Image(a), 1: orc.Image(b)}).slot_data(out, 0).image.planes
    assert_planes(whole, want, what="resizing Graph node vs oracle")
    # the second output of the same Graph node
    tp, lg = build(kc, g.dict(), {0: a, 1: b})
    full = lg.await_clean(gn).slot_data(gn, o2).image.planes()
    tp2, lg2 = build(kc, g.dict(), {0: a, 1: b})
    halves = [lg2.evaluate_band(gn, y0, y1, slot_id=o2).planes()[0] for (y0, y1) in ((0, 20), (20, 48))]
    assert_planes([np.concatenate(halves, axis=0)], full, what="second output slot by bands")


def test_bands_through_nested_graph_nodes(kc):
    from oracle import oracle as orc
    inner, inp, outn = _inner_invert()
    mid = G()
    m_in = mid.add({"InputGray": "x"})
    m_g = mid.add({"Graph": inner})
    m_mul = mid.add({"Mix": "Multiply"})
    m_out = mid.add({"OutputGray": "y"})
    mid.connect(m_in, m_g, 0, inp)
    mid.connect(m_g, m_mul, outn, 0)
    mid.connect(m_in, m_mul, 0, 1)
    mid.connect(m_mul, m_out, 0, 0)
    g = G()
    src = g.add({"Embed": 0})
    sep = g.add("SeparateRgba")
    gn = g.add({"Graph": mid.dict()})
    out = g.add({"OutputGray": "out"})
    g.connect(src, sep, 0, 0)
    g.connect(sep, gn, 2, m_in)
    g.connect(gn, out, m_out, 0)
    planes = [splitmix_plane(SEED_B, c, 29, 40) for c in range(4)]
    whole = check_bands(kc, g.dict(), {0: planes}, out, "nested Graph nodes")
    want = orc.RefGraph(g.dict(), embedded={0: orc.Image(planes)}).slot_data(out, 0).image.planes
    assert_planes(whole, want, what="nested Graph nodes vs oracle")
    assert_planes(whole, [(np.float32(1.0) - planes[2]) * planes[2]], what="closed form (1 - b) * b")
