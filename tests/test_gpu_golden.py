"""GPU parity, level 1: the reference's golden-image tests (tests/integration_tests.rs) run on
the HIP backend through the C ABI.  u8 output must equal the reference's golden PNG exactly
(`images_equal`, :38-45) and the f32 planes must equal the CPU oracle's (bit-exact; Pow <= 1 ulp)."""
import json
import os

import numpy as np
import pytest

from golden_graphs import COMPARE, GOLDEN_CASES, INPUTS, RESIZE_POLICY_CASES, resize_policy
from pngio import read_png
from util import assert_planes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    return kc


def _run(kc, graph, use_cache=False):
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    lg.use_cache = use_cache
    lg.set_base_dir(INPUTS)
    lg.set_node_graph(kc.NodeGraph.from_json(json.dumps(graph)))
    return lg


@pytest.mark.parametrize("use_cache", [False, True])
@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_golden_u8_and_oracle_f32(kc, name, use_cache, load_image):
    from oracle import oracle as orc
    graph, node, golden = GOLDEN_CASES[name]
    lg = _run(kc, graph, use_cache)
    got_u8 = lg.await_clean(node).buffer_rgba(node, 0)
    ref = orc.RefGraph(graph, load_image)
    want_img = ref.slot_data(node, 0).image
    got_img = lg.slot_data(node, 0).image
    assert got_img.is_rgba() == want_img.is_rgba
    # Pow is computed in f64 and rounded once: within 1 ulp of libm's powf
    ulp = 1 if name.startswith("pow") else 0
    assert_planes(got_img.planes(), want_img.planes, ulp=ulp, what=name)
    want_u8 = read_png(os.path.join(COMPARE, golden))
    assert got_u8.shape == want_u8.shape
    nbad = int((got_u8 != want_u8).sum())
    if name.startswith("pow"):
        # a 1-ulp difference may flip a truncation at an exact k/255 boundary
        assert nbad <= 8 and np.abs(got_u8.astype(int) - want_u8.astype(int)).max() <= 1, nbad
    else:
        assert nbad == 0, "%d of %d samples differ from the reference golden" % (nbad, want_u8.size)


@pytest.mark.parametrize("policy,img1,img2,expected", RESIZE_POLICY_CASES)
def test_resize_policy_sizes(kc, policy, img1, img2, expected):
    graph, mix = resize_policy(policy, img1, img2)
    lg = _run(kc, graph)
    assert lg.await_clean(mix).slot_data_size(mix, 0) == expected
