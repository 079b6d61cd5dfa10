"""GPU parity at BASELINE.json's full sizes: against the oracle where it finishes in seconds,
and through size-independent properties (fused == unfused bit for bit, crop equivalence of
pointwise graphs, alpha == 1, resize idempotence on constants) where it does not."""
import numpy as np
import pytest

from util import SEED_A, SEED_B, assert_planes, bit_equal, splitmix_plane

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    return kc


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    orc.set_threads(8)
    yield orc
    orc.set_threads(1)


@pytest.fixture(scope="module")
def planes4096():
    S = 4096
    a = [splitmix_plane(SEED_A, c, S, S) for c in range(3)]
    b = [splitmix_plane(SEED_B, c, S, S) for c in range(3)]
    one = np.ones((S, S), np.float32)
    return a + [one], b + [one]


def chain(kc, x, bb, w, h, n):
    white = kc.combine_rgba_process([kc.value_process(1.0)] * 3 + [None])
    for i in range(1, n + 1):
        if i & 1:
            x = kc.mix_process(x, bb, kc.MixType.Multiply if (i >> 1) & 1 else kc.MixType.Add)
        else:
            x = kc.mix_process(kc.resize_image(white, (w, h)), x, kc.MixType.Subtract)
    return x


@pytest.mark.parametrize("op", ["Add", "Divide"])
def test_config1_single_mix_4096_vs_oracle(kc, orc, planes4096, op):
    """BASELINE config #1: one Mix node on two 4096x4096 f32x4 inputs."""
    a, b = planes4096
    got = kc.mix_process(kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b), kc.MixType.parse(op)).planes()
    want = [orc.mix_plane(op, a[c], b[c]) for c in range(3)] + [np.ones_like(a[0])]
    assert_planes(got, want, what="4096 " + op)


def test_config3_chain32_4096_fused_equals_unfused_equals_oracle(kc, orc, planes4096):
    """BASELINE headline: 32-node linear mix/invert graph at 4096x4096."""
    a, b = planes4096
    S = 4096
    ia, ib = kc.SlotImage.from_planes(a), kc.SlotImage.from_planes(b)
    fused = chain(kc, ia, ib, S, S, 32).planes()
    kc.set_fusion(False)
    try:
        unfused = chain(kc, ia, ib, S, S, 32).planes()
    finally:
        kc.set_fusion(True)
    assert_planes(fused, unfused, what="fused vs unfused")
    assert bit_equal(fused[3], np.ones((S, S), np.float32))
    want = orc.chain32(a, b, 32)
    assert_planes(fused, want, what="fused vs oracle")


def test_config3_chain32_8192_crop_equivalence(kc, orc):
    """8192x8192 (BASELINE config #3 size): a pointwise graph commutes with cropping, so the
    oracle on a 192x160 crop of the inputs must equal the same crop of the GPU's full result."""
    S = 8192
    a = [splitmix_plane(SEED_A, c, S, S) for c in range(3)]
    b = [splitmix_plane(SEED_B, c, S, S) for c in range(3)]
    one = kc.SlotImage.from_value((S, S), 1.0, False).plane_handles  # noqa: F841 (keeps API exercised)
    ia = kc.combine_rgba_process([kc.SlotImage.from_planes([p]) for p in a] + [None])
    ib = kc.combine_rgba_process([kc.SlotImage.from_planes([p]) for p in b] + [None])
    got = chain(kc, ia, ib, S, S, 32).planes()
    for (y, x) in ((0, 0), (4000, 5000), (S - 160, S - 192)):
        ca = [p[y:y + 160, x:x + 192].copy() for p in a] + [np.ones((160, 192), np.float32)]
        cb = [p[y:y + 160, x:x + 192].copy() for p in b] + [np.ones((160, 192), np.float32)]
        want = orc.chain32(ca, cb, 32)
        assert_planes([p[y:y + 160, x:x + 192] for p in got], want, what="crop %d,%d" % (y, x))


def test_config3_chain32_8192_by_row_bands_vs_oracle_crops(kc, orc):
    """BASELINE config #3's actual multi-GPU shape: the 32-node graph at 8192x8192 split into two row bands, each evaluated
    through the band path (kc_live_graph_evaluate_band) on a graph that holds only that band's rows of the inputs -- what
    `bench.py --gpus 2` does on two GPUs, here one band after the other.  Three crops per band against the oracle."""
    from bench import add_chain
    from util import splitmix_rows
    S = 8192
    for (y0, y1) in ((0, S // 2), (S // 2, S)):
        rows = y1 - y0
        a = [splitmix_rows(SEED_A, c, S, S, y0, y1) for c in range(4)]
        b = [splitmix_rows(SEED_B, c, S, S, y0, y1) for c in range(4)]
        tp = kc.TextureProcessor.new()
        lg = tp.new_live_graph()
        lg.embed_slot_data_band(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 0, y0, S)
        lg.embed_slot_data_band(kc.SlotData(0, 0, kc.SlotImage.from_planes(b)), 1, y0, S)
        na, nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(0))), lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
        _, last = add_chain(kc, lg, na, nb, 32)
        got = lg.evaluate_band(last, y0, y1).planes()
        assert got[0].shape == (rows, S)
        for r0 in (0, rows // 2 - 16, rows - 32):
            for x0 in (0, S - 256):
                ca = [p[r0:r0 + 32, x0:x0 + 256].copy() for p in a]
                cb = [p[r0:r0 + 32, x0:x0 + 256].copy() for p in b]
                want = orc.chain32(ca, cb, 32)
                assert_planes([p[r0:r0 + 32, x0:x0 + 256] for p in got], want, what="band %d:%d crop %d,%d" % (y0, y1, r0, x0))
        del got, lg, tp, a, b


def test_config2_resize_512_to_4096_and_blend_chain_vs_oracle(kc, orc, planes4096):
    """BASELINE config #2: B 512^2 -> 4096^2 (Triangle, MostPixels) feeding a 3-node blend chain,
    through the LiveGraph so the implicit resize pre-step runs per consuming node."""
    a, _ = planes4096
    S, s = 4096, 512
    b = [splitmix_plane(SEED_B, c, s, s) for c in range(4)]
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 0)
    lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(b)), 1)
    na = lg.add_node(kc.Node.new(kc.NodeType.Embed(0)))
    nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
    n1 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Add)))
    n2 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Multiply)))
    n3 = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.Subtract)))
    lg.connect(na, n1, 0, 0)
    lg.connect(nb, n1, 0, 1)
    lg.connect(n1, n2, 0, 0)
    lg.connect(na, n2, 0, 1)
    lg.connect(n2, n3, 0, 0)
    lg.connect(nb, n3, 0, 1)
    got = lg.await_clean(n3).slot_data(n3, 0).image.planes()
    bu = [orc.resize_plane(p, S, S, "Triangle") for p in b[:3]]
    want = []
    for c in range(3):
        x1 = orc.mix_plane("Add", a[c], bu[c])
        x2 = orc.mix_plane("Multiply", x1, a[c])
        want.append(orc.mix_plane("Subtract", x2, bu[c]))
    want.append(np.ones((S, S), np.float32))
    assert_planes(got, want, what="config #2")
    assert lg.slot_data_size(n3, 0) == (S, S)


def test_height_to_normal_4096_vs_oracle(kc, orc):
    p = splitmix_plane(SEED_A, 1, 4096, 4096)
    got = kc.height_to_normal_process(kc.SlotImage.from_planes([p])).planes()
    nx, ny, nz = orc.height_to_normal(p)
    assert_planes(got, [nx, ny, nz, np.ones_like(p)], what="h2n 4096")


@pytest.mark.parametrize("srgb", [False, True])
def test_to_u8_4096_vs_oracle(kc, orc, planes4096, srgb):
    a, _ = planes4096
    got = kc.SlotImage.from_planes(a).to_u8(srgb)
    assert np.array_equal(got, orc.to_u8(orc.Image(a), srgb))


# Down-sampling on both axes at full size (resize_down_kernel: ~1 200 tiles, partial last tiles in both directions,
# non-integer ratios, widths that are not multiples of 4), with non-finite samples that must stay inside their windows.
@pytest.mark.parametrize("filt,src,dst", [
    ("Lanczos3", (4096, 4096), (1024, 1024)),
    ("Gaussian", (3000, 3000), (700, 700)),
    ("CatmullRom", (4096, 4096), (1365, 1365)),
    ("Triangle", (4093, 2050), (511, 259)),
    ("Lanczos3", (4096, 4096), (3000, 3000)),
])
def test_resize_down_full_size_vs_oracle(kc, orc, filt, src, dst):
    (sw, sh), (dw, dh) = src, dst
    p = splitmix_plane(SEED_B, 1, sh, sw) * np.float32(1.5) - np.float32(0.25)  # exercises the [0, 1] clamp
    p[sh // 2, sw // 2 - 2:sw // 2 + 2] = [np.nan, np.inf, -np.inf, -0.0]
    p[0, 0] = np.inf
    p[-1, -1] = -np.inf
    got = kc.resize_image(kc.SlotImage.from_planes([p]), (dw, dh), kc.ResizeFilter.parse(filt)).planes()[0]
    want = orc.resize_plane(p, dw, dh, filt)
    assert bit_equal(got, want), "%s %s->%s" % (filt, src, dst)
    # the non-finite samples reach exactly the outputs whose windows contain them
    assert np.isnan(got).sum() == np.isnan(want).sum() > 0


@pytest.mark.parametrize("size", [4096, 8192])
def test_compiled_chain_kernels_at_full_size_vs_oracle_crops(kc, orc, size):
    """The kernels compiled at run time (csrc/specialize.cpp, "compile at first sight") at BASELINE's full sizes: their index
    arithmetic runs over 16.7 M / 67 M float4 units here, where the small-size tests of tests/test_gpu_specialize.py cannot see
    it.  The 32-node graph through the LiveGraph (one launch of a kc_chain_* kernel), the first and the last rows and columns and
    a band in the middle against the oracle on crops; then config #4's 16-plane program at 4096^2 the same way."""
    from bench import add_chain
    S = size
    a = [splitmix_plane(SEED_A, c, S, S) for c in range(4)]
    b = [splitmix_plane(SEED_B, c, S, S) for c in range(4)]
    mode = kc.get_specialize()
    kc.set_specialize(2)
    try:
        s0 = kc.specialize_stats()
        tp = kc.TextureProcessor.new()
        lg = tp.new_live_graph()
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(a)), 0)
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(b)), 1)
        na, nb = lg.add_node(kc.Node.new(kc.NodeType.Embed(0))), lg.add_node(kc.Node.new(kc.NodeType.Embed(1)))
        _, last = add_chain(kc, lg, na, nb, 32)
        got = lg.await_clean(last).slot_data(last, 0).image.planes()
        s1 = kc.specialize_stats()
        assert s1["specialized_launches"] == s0["specialized_launches"] + 1, "the evaluation did not run a compiled kernel"
        crops = [(0, 0), (S // 2 - 16, S // 2 - 128), (S - 32, S - 256), (0, S - 256), (S - 32, 0)]
        for (y, x) in crops:
            ca = [p[y:y + 32, x:x + 256].copy() for p in a]
            cb = [p[y:y + 32, x:x + 256].copy() for p in b]
            assert_planes([p[y:y + 32, x:x + 256] for p in got], orc.chain32(ca, cb, 32), what="%d^2 crop %d,%d" % (S, y, x))
        del got, lg, tp
        if S == 4096:
            # config #4 on one GPU: 8 branches + add tree = ONE launch of a program that reads 16 planes per channel
            from rank_scenarios import config4_graph
            import json
            graph, root = config4_graph()
            imgs = {e: [splitmix_plane(0x5EED0100 + e, c, S, S) for c in range(4)] for e in range(16)}
            tp = kc.TextureProcessor.new()
            lg = tp.new_live_graph()
            lg.set_node_graph(kc.NodeGraph.from_json(json.dumps(graph)))
            for e, planes in imgs.items():
                lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(planes)), e)
            n0 = kc.stats_counter("wide_launches")
            got = lg.await_clean(root).slot_data(root, 0).image.planes()
            assert kc.stats_counter("wide_launches") == n0 + 1
            for (y, x) in crops:
                parts = [orc.chain32([p[y:y + 32, x:x + 256].copy() for p in imgs[2 * k]], [p[y:y + 32, x:x + 256].copy() for p in imgs[2 * k + 1]], 16)[:3]
                         for k in range(8)]
                while len(parts) > 1:
                    parts = [[orc.mix_plane("Add", parts[i][c], parts[i + 1][c]) for c in range(3)] for i in range(0, len(parts), 2)]
                assert_planes([p[y:y + 32, x:x + 256] for p in got[:3]], parts[0], what="config #4 4096^2 crop %d,%d" % (y, x))
    finally:
        kc.set_specialize(mode, 2)
