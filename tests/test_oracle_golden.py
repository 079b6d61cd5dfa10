"""Pins the CPU oracle: every golden PNG the reference's tests compare against
(tests/integration_tests.rs, data/test_compare/*.png) must be reproduced bit-exactly at u8 by
oracle.RefGraph -- the same check the reference's `images_equal` makes (:38-45)."""
import os

import numpy as np
import pytest

from golden_graphs import COMPARE, GOLDEN_CASES, RESIZE_POLICY_CASES, resize_policy
from oracle import oracle as orc
from pngio import read_png


@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_oracle_reproduces_reference_golden(name, load_image):
    graph, node, golden = GOLDEN_CASES[name]
    got = orc.RefGraph(graph, load_image).buffer_rgba(node, 0)
    want = read_png(os.path.join(COMPARE, golden))
    assert got.shape == want.shape
    assert np.array_equal(got, want), "%d of %d samples differ" % ((got != want).sum(), want.size)


@pytest.mark.parametrize("policy,img1,img2,expected", RESIZE_POLICY_CASES)
def test_oracle_resize_policy_sizes(policy, img1, img2, expected, load_image):
    # tests/integration_tests.rs:894-949
    graph, mix = resize_policy(policy, img1, img2)
    assert orc.RefGraph(graph, load_image).slot_data(mix, 0).size == expected


def test_oracle_numeric_kats():
    # read_dirty_read (tests/integration_tests.rs:1388-1437): Value 0.5 -> Combine -> [127, 0, 0, 255]
    g = {"nodes": [{"node_id": 0, "node_type": {"Value": 0.5}}, {"node_id": 1, "node_type": "CombineRgba"}],
         "edges": [{"output_id": 0, "input_id": 1, "output_slot": 0, "input_slot": 0}]}
    assert orc.RefGraph(g).buffer_rgba(1, 0).reshape(-1).tolist() == [127, 0, 0, 255]
    # drive_cache (:143,225): exact f32 values through Combine
    vals = [0.0, 0.3, 0.7, 1.0]
    g = {"nodes": [{"node_id": i, "node_type": {"Value": v}} for i, v in enumerate(vals)]
         + [{"node_id": 4, "node_type": "CombineRgba"}],
         "edges": [{"output_id": i, "input_id": 4, "output_slot": 0, "input_slot": i} for i in range(4)]}
    img = orc.RefGraph(g).slot_data(4, 0).image
    assert [float(p[0, 0]) for p in img.planes] == [float(np.float32(v)) for v in vals]
    # request_empty_buffer (:307-333): Mix with no inputs -> 1x1 gray 0.0 -> [0,0,0,255]
    g = {"nodes": [{"node_id": 0, "node_type": {"Mix": "Add"}}, {"node_id": 1, "node_type": {"OutputRgba": "out"}}],
         "edges": [{"output_id": 0, "input_id": 1, "output_slot": 0, "input_slot": 0}]}
    assert orc.RefGraph(g).buffer_rgba(1, 0).reshape(-1).tolist() == [0, 0, 0, 255]


def test_to_u8_edge_cases():
    # SlotImage::f32_to_u8 (src/slot_image.rs:141-144): truncation, NaN -> 255, +-Inf, -0.0
    v = np.array([[0.0, -0.0, 1.0, 2.0, -1.0, np.inf, -np.inf, np.nan, 0.5, 0.999999, 1e-40, 254.5 / 255]], np.float32)
    got = orc.to_u8(orc.Image([v]))[0, :, 0].tolist()
    assert got == [0, 0, 255, 255, 0, 255, 0, 255, 127, 254, 0, 254]


def test_weak_legacy_downsample_evidence(load_image):
    # Unreferenced legacy golden: Mix(Add)(heart_128, Triangle 256->128 of
    # heart_256) matches resize_policy_least_pixels.png except <= 1 LSB in a handful of pixels.
    graph, mix = resize_policy("LeastPixels", "heart_128.png", "heart_256.png")
    got = orc.RefGraph(graph, load_image).buffer_rgba(mix, 0).astype(int)
    want = read_png(os.path.join(COMPARE, "resize_policy_least_pixels.png")).astype(int)
    diff = np.abs(got - want)
    assert diff.max() <= 1 and (diff != 0).sum() <= 4
