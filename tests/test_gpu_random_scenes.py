"""Randomised differential test of the HIP path against the CPU oracle (tools/diff_fuzz.py): 180 seeded random scenes -
degenerate triangles, integer grids, flat and tiny clusters, coordinates outside the fast-division ranges - with random
rays, bit for bit under both visiting rules, and 45 of them as frames under rotating feature sets."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_scenes_match_the_oracle(orc):
    import diff_fuzz
    assert diff_fuzz.run(10000, 180) == 0


def test_random_soups_build_like_the_host_builder():
    """Random triangle soups (ties, sorted, reversed, clustered, all-equal keys; 1 to 60 000 triangles): the device-side
    build gives the host builder's arrays bit for bit."""
    import build_fuzz
    assert build_fuzz.run(5000, 120) == 0
