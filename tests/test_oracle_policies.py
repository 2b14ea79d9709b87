"""The traversal policies of the CPU restatement against each other (no GPU): skipping the boxes a ray misses
(F_MISS_CULL) and testing shadow queries on outward-rounded 16-bit boxes (F_COARSE_SHADOW) are ways of visiting fewer
nodes, not of answering differently, so on the prebuilt scenes they give the frames of the reference's own visiting rule
up to the rounding of hit points, and each other's bit for bit."""
import numpy as np
import pytest


@pytest.mark.parametrize("name,res", [("cornell_parallelogram", (40, 40)), ("monkey", (64, 48)), ("cube", (48, 36)), ("teapot", (48, 27))])
def test_coarse_shadow_boxes_answer_like_exact_ones(orc, scenes, name, res):
    b = orc.Bvh(scenes(name), orc.F_ACCEL)
    cam = orc.make_camera(*res)
    f = orc.F_ACCEL | orc.F_SHADING | orc.F_HARD_SHADOW | orc.F_SOFT_SHADOW | orc.F_RECURSIVE | orc.F_MISS_CULL
    exact = b.render(cam, f, seed=3)
    coarse = b.render(cam, f | orc.F_COARSE_SHADOW, seed=3)
    assert exact["counters"]["shadow_rays"] == coarse["counters"]["shadow_rays"] > 0
    assert np.array_equal(exact["rgb"].view(np.uint32), coarse["rgb"].view(np.uint32))
    assert exact["rgb"].max() > 0.05


def test_policies_against_the_reference_rule(orc, scenes):
    b = orc.Bvh(scenes("monkey"), orc.F_ACCEL)
    cam = orc.make_camera(48, 36)
    f = orc.F_ACCEL | orc.F_SHADING | orc.F_HARD_SHADOW | orc.F_RECURSIVE
    ref = b.render(cam, f)
    for policy in (orc.F_MISS_CULL, orc.F_MISS_CULL | orc.F_COARSE_SHADOW):
        got = b.render(cam, f | policy)
        assert np.array_equal(got["tri"], ref["tri"])
        assert np.abs(got["rgb"] - ref["rgb"]).max() <= 1.0 / 255.0
        assert got["counters"]["inner_visits"] < ref["counters"]["inner_visits"]


def test_coarse_flag_needs_miss_culling(orc, scenes):
    """Without F_MISS_CULL the flag changes nothing: the reference rule tests no shadow box before the first hit."""
    b = orc.Bvh(scenes("cornell"), orc.F_ACCEL)
    cam = orc.make_camera(32, 32)
    f = orc.F_ACCEL | orc.F_SHADING | orc.F_HARD_SHADOW
    a, c = b.render(cam, f), b.render(cam, f | orc.F_COARSE_SHADOW)
    assert np.array_equal(a["rgb"].view(np.uint32), c["rgb"].view(np.uint32)) and a["counters"] == c["counters"]
