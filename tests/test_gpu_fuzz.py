"""Randomised differential test: scenes built straight into the reference's buffer layouts — triangle soups from 1e-2 to
100 units, slivers, exact duplicates and coplanar stacks (dst ties), integer-grid quads (rays through shared edges), zero-area,
NaN, infinite and 1e38-sized triangles, both RNG modes, chunk boxes that are tight (the reference's), loose or too tight (FLAT_CHUNKS must cut the same triangles
off), spheres around the camera — with random tracer settings and random tuning knobs (stack spill, loop thresholds, leaf size, builder).  GPU (each kernel in turn) == oracle, bit for bit, and the
oracle's search tree == its literal loop on the same input."""
import os

import numpy as np
import pytest

from test_gpu_parity import assert_bitwise, run_gpu

pytestmark = pytest.mark.gpu


def _unit(v):
    with np.errstate(invalid="ignore", over="ignore"):
        n = np.linalg.norm(v, axis=-1, keepdims=True)
        return np.where(n > 0, v / np.where(n > 0, n, 1), 0).astype(np.float32)


def random_scene(rtx, seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = (64, 48) if seed % 5 else (int(rng.integers(1, 90)), int(rng.integers(1, 70)))      # every fifth: odd sizes
    p, _, _, _ = rtx.scenes.mesh_test_scene(w, h).build_buffers()
    p = p.copy()
    p["maxBounceCount"] = int(rng.integers(0, 7)) if seed % 7 else int(rng.integers(7, 14))
    p["numRaysPerPixel"] = int(rng.integers(1, 5))
    if seed % 4 == 3:                             # every fourth: the counter-based mode (its own estimator tree: 1, 4 or 16 sample lanes per pixel)
        p["rngMode"] = 1
        p["numRaysPerPixel"] = int(np.random.default_rng(77 + seed).choice([1, 3, 4, 6, 16, 19]))
    far = seed % 11 == 10                         # every eleventh: the whole scene 1e5 units away from the origin
    shift = np.float32([1e5, -2e5, 5e4]) if far else np.float32([0, 0, 0])
    if far:
        for k, a in enumerate((3, 7, 11)):
            p["camLocalToWorld"][a] += shift[k]
        p["worldSpaceCameraPos"] = p["worldSpaceCameraPos"] + shift
    p["defocusStrength"] = float(rng.choice([0.0, 0.0, 30.0, 200.0]))
    p["divergeStrength"] = float(rng.choice([0.0, 0.3, 2.0]))
    p["environmentEnabled"] = int(rng.integers(0, 2))
    p["sunIntensity"] = float(rng.choice([0.0, 10.0]))
    p["intersectMode"] = int(rng.integers(0, 2))
    chunks = []                                   # list of (positions [n,3,3] float32, box policy)
    for _ in range(int(rng.integers(3, 9))):
        kind = rng.choice(["soup", "soup", "sliver", "stack", "grid", "big", "degenerate"])
        centre = rng.uniform([-4, 0, -3], [4, 3, 5]).astype(np.float32)
        if kind == "soup":
            s = float(10.0 ** rng.uniform(-2, 0.7))
            pos = centre + rng.uniform(-s, s, (int(rng.integers(1, 60)), 3, 3))
        elif kind == "sliver":
            n = int(rng.integers(1, 20))
            a = centre + rng.uniform(-2, 2, (n, 1, 3))
            pos = np.concatenate([a, a + rng.uniform(-2, 2, (n, 1, 3)), a + rng.uniform(-1e-4, 1e-4, (n, 1, 3))], axis=1)
        elif kind == "stack":                     # the same triangles several times, plus coplanar overlapping ones
            base = centre + rng.uniform(-1.5, 1.5, (int(rng.integers(1, 6)), 3, 3))
            base[:, :, 2] = centre[2]             # all in the plane z = const
            pos = np.concatenate([base, base, base[::-1], base + np.float32([0.25, 0.0, 0.0])])
        elif kind == "grid":
            k = int(rng.integers(2, 6)); y = float(np.floor(centre[1]))
            q = []
            for i in range(k):
                for j in range(k):
                    a, b, c, d = (i, y, j), (i, y, j + 1), (i + 1, y, j + 1), (i + 1, y, j)
                    q += [[a, b, c], [a, c, d]]
            pos = np.array(q, np.float32) + np.float32([np.floor(centre[0]) - k / 2, 0, np.floor(centre[2])])
        elif kind == "big":
            pos = centre + rng.uniform(-100, 100, (2, 3, 3))
        else:
            pos = centre + rng.uniform(-1, 1, (4, 3, 3))
            pos[0, 1] = pos[0, 0]                 # zero area
            pos[1, 2] = pos[1, 1] = pos[1, 0]     # a point
            if seed % 4 == 0:
                pos[2, 2, 0] = np.nan
            elif seed % 4 == 1:
                pos[3, 0, 1] = np.inf
            elif seed % 4 == 2:
                pos[3] = pos[3] * np.float32(3e37)        # products overflow to inf inside RayTriangle
        chunks.append(((pos + shift).astype(np.float32), rng.choice(["tight", "tight", "tight", "loose", "cut"])))
    n = sum(len(c[0]) for c in chunks)
    tris = np.zeros(n, rtx.TRIANGLE)
    infos = np.zeros(len(chunks), rtx.MESHINFO)
    at = 0
    for ci, (pos, policy) in enumerate(chunks):
        k = len(pos)
        t = tris[at:at + k]
        t["posA"], t["posB"], t["posC"] = pos[:, 0], pos[:, 1], pos[:, 2]
        with np.errstate(invalid="ignore", over="ignore"):
            face = _unit(np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]))
        for f in ("normalA", "normalB", "normalC"):
            t[f] = _unit(face + rng.uniform(-0.3, 0.3, (k, 3)).astype(np.float32)) if rng.random() < 0.8 else 0.0
        mi = infos[ci]
        mi["firstTriangleIndex"], mi["numTriangles"] = at, k
        mat = mi["material"]
        mat["colour"] = (*rng.uniform(0.2, 1, 3), 1); mat["emissionColour"] = (*rng.uniform(0, 1, 3), 1)
        mat["specularColour"] = (1, 1, 1, 1)
        mat["emissionStrength"] = float(rng.choice([0, 0, 3]))
        mat["smoothness"] = float(rng.choice([0, 0.5, 1])); mat["specularProbability"] = float(rng.choice([0, 0.3, 1]))
        mat["flag"] = int(rng.choice([0, 0, 1, 2]))
        with np.errstate(invalid="ignore", over="ignore"):
            lo, hi = np.nanmin(pos.reshape(-1, 3), axis=0), np.nanmax(pos.reshape(-1, 3), axis=0)
            if policy == "loose":
                lo, hi = lo - 0.5, hi + 0.5
            elif policy == "cut":
                mid = 0.5 * (lo + hi); lo, hi = mid - 0.3 * (hi - lo), mid + 0.3 * (hi - lo)
        mi["boundsMin"], mi["boundsMax"] = lo, hi
        at += k
    ns = int(rng.integers(0, 5))
    sph = np.zeros(ns, rtx.SPHERE)
    for s in sph:
        s["position"] = rng.uniform([-4, 0, -8], [4, 3, 4]) + shift
        s["radius"] = float(10.0 ** rng.uniform(-1, 0.6)) if rng.random() < 0.9 else 0.0
        s["material"]["colour"] = (*rng.uniform(0.2, 1, 3), 1); s["material"]["specularColour"] = (1, 1, 1, 1)
        s["material"]["emissionStrength"] = float(rng.choice([0, 2])); s["material"]["emissionColour"] = (1, 1, 1, 1)
        s["material"]["smoothness"] = float(rng.choice([0, 1])); s["material"]["specularProbability"] = float(rng.choice([0, 1]))
        s["material"]["flag"] = int(rng.choice([0, 1, 2]))
    if seed % 3 == 2:
        # AllMeshInfo in a different order than the triangle buffer (non-monotone firstTriangleIndex): the reference's loop visits
        # chunks in list order, so equal-distance hits (stacks, duplicates) go to the chunk that comes first in the LIST
        infos = infos[rng.permutation(len(infos))].copy()
    if seed % 6 == 5 and ns > 0:
        # a mirror sphere ~1e3 units from the unit-scale meshes: its bounce rays start far outside the triangles' extent
        sph[0]["position"] = np.float32([700.0, 650.0, -300.0]) + shift
        sph[0]["radius"] = 400.0
        sph[0]["material"]["smoothness"] = 1.0; sph[0]["material"]["specularProbability"] = 1.0; sph[0]["material"]["flag"] = 0
    return p, sph, tris, infos


_FIRST = int(os.environ.get("RTX_FUZZ_FIRST", "0"))
@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + int(os.environ.get("RTX_FUZZ_SEEDS", "24"))))   # RTX_FUZZ_SEEDS=400 [RTX_FUZZ_FIRST=300] for a soak run
def test_random_scene_matches_the_oracle(rtx, oracle, tracer, seed):
    b = random_scene(rtx, seed)
    kernel = (0, 1, 1, 1, 0, 1, -1)[seed % 7]
    rng = np.random.default_rng(seed)
    knobs = {"stream_stack": int(rng.choice([4, 9, 30, 37])), "node_min": int(rng.choice([1, 6, 24, 64])), "tiles_per_fetch": int(rng.choice([1, 2, 5, 40])), "fetch_guide": int(rng.choice([1, 4, 16])),
             "max_leaf": int(rng.choice([1, 2, 4])), "full_sort": int(rng.integers(0, 2)), "frame_batch": int(rng.choice([0, 1])),
             "bvh_reinsert": int(rng.choice([0, 0, 2])),
             "stream_tile": int(rng.choice([0, 2, 4])), "compact_nodes": int(rng.choice([0, 1, 1])), "tile_lpt": int(rng.choice([0, 1, 1])),
             "device_bvh": int(rng.choice([0, 1])), "bvh_collapse": int(rng.choice([0, 1, 2])), "bvh_radius": int(rng.choice([2, 8, -16, 40])), "bvh_top": int(rng.choice([0, 2, 600, 1024])), "bvh_treelets": int(rng.choice([0, 1, 3, 6]))}
    defaults = {"stream_stack": 0, "node_min": 10, "tiles_per_fetch": 16, "max_leaf": 2, "full_sort": 0, "frame_batch": 0,
                "bvh_reinsert": 0, "fetch_guide": 4, "stream_tile": 4, "compact_nodes": 1, "tile_lpt": 1, "device_bvh": -1, "bvh_collapse": 0, "bvh_radius": -16, "bvh_top": 0, "bvh_treelets": 6}
    nf = int(rng.choice([2, 2, 4, 5]))                  # 4 and 5: whole frame groups of 4 (+ a remainder) when stream_tile allows
    for k, v in knobs.items():
        tracer.set_option(k, v)
    try:
        acc, last = run_gpu(tracer, b, seed, nf, kernel=kernel, shade_threshold=int(rng.choice([1, 24, 48, 64])))
        rays = tracer.stats()["rays"]
    finally:
        for k, v in defaults.items():
            tracer.set_option(k, v)
    want_acc, want_last, cnt = oracle.render(*b, seed, nf, accel=True)
    what = f"fuzz seed {seed} (kernel {kernel}, {len(b[2])} triangles in {len(b[3])} chunks, {len(b[1])} spheres, mode {int(b[0]['intersectMode'])})"
    assert_bitwise(last, want_last, what + ", last frame")
    assert_bitwise(acc, want_acc, what + ", accum")
    assert rays == cnt["rays"]
    loop_acc, _, lc = oracle.render(*b, seed, nf)
    assert_bitwise(want_acc, loop_acc, what + ": oracle tree vs oracle loop")
    assert lc["rays"] == cnt["rays"]
