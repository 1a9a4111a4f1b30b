"""The host BVH builder (csrc/bvh.cpp) on the CPU: for every combination of collapse (greedy, cost-driven with its own leaves, cost-driven
over the split search's leaves), leaf size and insertion passes, every triangle must sit in exactly one leaf and inside every box on the
way down to it.  (Round 4: the randomised GPU test met a scene where the cost-driven collapse after insertion passes lost three
triangles — the collapse took a subtree's triangles as one index range, which the insertion passes had broken up.)"""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray-tracing-extended_amd", "csrc")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bvh") / "bvh_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", CSRC, os.path.join(ROOT, "tests", "bvh_check.cpp"), os.path.join(CSRC, "bvh.cpp"), "-o", exe])
    return exe


def _soup(seed, n):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-5, 5, (n, 1, 3)).astype(np.float32)
    size = np.float32(10.0) ** rng.uniform(-2, 0.5, (n, 1, 1)).astype(np.float32)
    p = c + rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32) * size
    if n >= 8:
        p[3] = p[2]                                   # an exact duplicate
        p[5] = p[5, 0]                                # a point
        p[6, :, 1] = 0.0                              # axis-aligned, zero thickness
    if n >= 40:
        p[7] = np.float32([[-50, -1, -50], [50, -1, -50], [0, -1, 60]])      # a floor under everything
    return p.reshape(n, 9)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 109, 700, 5000])
def test_every_triangle_in_exactly_one_leaf_inside_its_boxes(checker, tmp_path, n):
    path = str(tmp_path / "t.f32")
    _soup(n, n).tofile(path)
    for collapse in (0, 1, 2):
        for max_leaf in (1, 2, 4):
            for passes in (0, 2):
                out = subprocess.run([checker, path, str(collapse), str(max_leaf), str(passes)], capture_output=True, text=True, timeout=120).stdout
                m = re.search(r"(\d+) triangles, (\d+) nodes: (\d+) missing, (\d+) duplicated, (\d+) containment errors", out)
                assert m, out
                assert (int(m.group(1)), int(m.group(3)), int(m.group(4)), int(m.group(5))) == (n, 0, 0, 0), (n, collapse, max_leaf, passes, out)
