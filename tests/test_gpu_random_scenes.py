"""Randomised differential test of the HIP path against the CPU oracle (tools/diff_fuzz.py): 180 seeded random scenes -
degenerate triangles, integer grids, flat and tiny clusters, coordinates outside the fast-division ranges - with random
rays, bit for bit under both visiting rules, and 45 of them as frames under rotating feature sets."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_scenes_match_the_oracle(orc):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import diff_fuzz
    assert diff_fuzz.run(10000, 180) == 0
