"""CPU suite, part 1: the oracle is pinned against every known answer the reference's own files provide, and against
the committed golden images (regenerated only by tests/golden/make_fixtures.py)."""
import ctypes
import json
import math
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_pcg_known_answers(oracle):
    """RayTracing.shader:193-199 — integer-exact KATs."""
    kat = json.load(open(os.path.join(GOLDEN, "pcg_kat.json")))
    assert kat["0"]["outputs"] == [129708002, 582399676, 1006035121, 1462727737] and kat["0"]["state"] == 878960812
    assert kat["2073599"]["outputs"] == [2921424543, 1954327279, 3785153123, 3903738793]
    for seed, exp in kat.items():
        s = ctypes.c_uint32(int(seed))
        got = [oracle.lib.orc_next_random(ctypes.byref(s)) for _ in range(4)]
        assert got == exp["outputs"] and s.value == exp["state"], seed


def test_random_value_is_r_times_2_pow_minus_32(oracle):
    """shader :203 — the literal 4294967295.0 is float32 2^32; uint->float is RNE; range [0, 1] inclusive."""
    for seed in (0, 1, 12345, 0xFFFFFFFF):
        s1, s2 = ctypes.c_uint32(seed), ctypes.c_uint32(seed)
        r = oracle.lib.orc_next_random(ctypes.byref(s1))
        v = oracle.lib.orc_random_value(ctypes.byref(s2))
        assert np.float32(v) == np.float32(np.float32(r) * np.float32(2.0 ** -32))
        assert 0.0 <= v <= 1.0


def _ulp_err(f, ref, xs):
    worst = 0.0
    for x in xs:
        x = float(np.float32(x))
        r = ref(x)
        if r == 0 or not math.isfinite(r):
            continue
        worst = max(worst, abs(f(x) - r) / abs(float(np.spacing(np.float32(r)))))
    return worst


def test_frozen_transcendentals_are_accurate(oracle):
    """The oracle's own sin/cos/log/exp2 (needed for bit-identical CPU/GPU results) stay within 2 ulp of libm on the
    ranges the shader feeds them."""
    L = oracle.lib
    rng = np.random.default_rng(0)
    ang = rng.uniform(0, 6.2832, 4000)
    assert _ulp_err(L.om_sin, math.sin, ang) < 2.0
    assert _ulp_err(L.om_cos, math.cos, ang) < 2.0
    u = np.concatenate([rng.uniform(0, 1, 4000), 2.0 ** rng.uniform(-32, 0, 1000), rng.uniform(0.9, 1.1, 1000)])
    assert _ulp_err(L.om_log, math.log, u) < 1.5
    assert _ulp_err(L.om_exp2, lambda x: 2.0 ** x, rng.uniform(-40, 10, 4000)) < 2.0
    assert L.om_log(0.0) == -math.inf and L.om_log(1.0) == 0.0
    assert L.om_pow(0.0, 0.35) == 0.0 and abs(L.om_pow(0.5, 0.35) - 0.5 ** 0.35) < 1e-6
    assert L.om_exp2(-200.0) == 0.0 and L.om_exp2(200.0) == math.inf
    assert L.om_cos(0.0) == 1.0 and L.om_sin(0.0) == 0.0


def test_min_max_return_the_non_nan_operand(oracle):
    L = oracle.lib
    nan = float("nan")
    assert L.om_max(nan, 0.0) == 0.0 and L.om_max(0.0, nan) == 0.0
    assert L.om_min(nan, 1.0) == 1.0 and L.om_min(1.0, nan) == 1.0
    assert math.isnan(L.om_min(nan, nan))


def test_accumulate_is_clamped_running_average(oracle):
    """Accumulate.shader:43-54 — weight 1/(frame+1), saturate each frame, NaN -> 0."""
    acc = np.zeros(8, np.float32)
    oracle.accumulate(acc, np.array([0.5, 2.0, -1.0, np.nan, 1.0, 0.25, 3.0, 0.0], np.float32), 0)
    assert acc.tolist() == [0.5, 1.0, 0.0, 0.0, 1.0, 0.25, 1.0, 0.0]
    oracle.accumulate(acc, np.array([1.0, 0.0, 1.0, 1.0, 1.0, 0.75, 0.0, 0.5], np.float32), 1)
    assert acc.tolist() == [0.75, 0.5, 0.5, 0.5, 1.0, 0.5, 0.5, 0.25]


def _assert_bits(a, b, what):
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{what}: {int((~same).any(-1).sum())} pixels differ"


def test_oracle_reproduces_config1_golden(rtx, oracle):
    """configs[0] in full (16 spheres, 256x256, 4 rays, 3 bounces, frame 0)."""
    z = np.load(os.path.join(GOLDEN, "config1_golden.npz"))
    acc, last, cnt = oracle.render(*rtx.scenes.config1().build_buffers(), 0, 1)
    _assert_bits(last, z["frame0"], "config1 frame")
    _assert_bits(acc, z["accum0"], "config1 accum")
    meta = json.loads(bytes(z["meta"]).decode())
    assert cnt["rays"] == meta["counts"]["rays"] and cnt["sphereTests"] == 16 * cnt["rays"]


@pytest.mark.parametrize("name", ["Chess", "Knight", "Reflective_Balls", "Balls_Outdoors"])
def test_oracle_reproduces_reference_scene_goldens(rtx, oracle, name):
    from rtx_amd import unity_scene
    z = np.load(os.path.join(GOLDEN, f"{name}_golden.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    m = unity_scene.load_scene_npz(os.path.join(GOLDEN, "scenes", name + ".npz"), meta["width"], meta["height"])
    m.numRaysPerPixel, m.maxBounceCount = meta["rays"], meta["bounces"]
    acc, last, cnt = oracle.render(*m.build_buffers(), 0, meta["frames"])
    _assert_bits(acc, z["accum"], name)
    assert cnt["rays"] == meta["counts"]["rays"] and cnt["triTests"] == meta["counts"]["triTests"]


def test_oracle_thread_count_does_not_change_bits(rtx, oracle):
    b = rtx.scenes.mesh_test_scene(48, 32).build_buffers()
    a, _ = oracle.render_frame(*b, 2, nthreads=1)
    c, _ = oracle.render_frame(*b, 2, nthreads=4)
    _assert_bits(a, c, "1 vs 4 threads")


def test_oracle_crop_equals_window_of_full_frame(rtx, oracle):
    b = rtx.scenes.mesh_test_scene(48, 32).build_buffers()
    full, _ = oracle.render_frame(*b, 1)
    crop, _ = oracle.render_frame(*b, 1, rect=(10, 5, 30, 25))
    _assert_bits(crop, full[5:25, 10:30], "crop")


def test_flat_chunks_vs_brute_on_reference_scene(rtx, oracle):
    """The chunk AABB cull (RayTracing.shader:279) is semantically a no-op; report where float rounding disagrees."""
    from rtx_amd import unity_scene
    m = unity_scene.load_scene_npz(os.path.join(GOLDEN, "scenes", "Knight.npz"), 64, 36)
    m.numRaysPerPixel, m.maxBounceCount = 2, 3
    b = m.build_buffers()
    flat, cf = oracle.render_frame(*b, 0, mode=0)
    brute, cb = oracle.render_frame(*b, 0, mode=1)
    differing = int(((flat.view(np.uint32) != brute.view(np.uint32)) & ~(np.isnan(flat) & np.isnan(brute))).any(-1).sum())
    assert differing <= 2, differing
    assert cf["boxTests"] == 39 * cf["rays"] and cb["boxTests"] == 0 and cb["triTests"] == 530 * cb["rays"]


@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_search_tree_returns_the_loops_image(rtx, oracle, mode):
    """orc_set_accel: the oracle's own search tree (used to check full-size frames) must give the literal loop's image
    bit for bit — spheres + meshes, the Knight scene (530-triangle chunks) and a window of the 100k-triangle scene."""
    from rtx_amd import unity_scene
    cases = [(rtx.scenes.mesh_test_scene(64, 40), None)]
    k = unity_scene.load_scene_npz(os.path.join(GOLDEN, "scenes", "Knight.npz"), 64, 36)
    k.numRaysPerPixel, k.maxBounceCount = 2, 3
    cases.append((k, None))
    c3 = rtx.scenes.config3(480, 270)
    c3.numRaysPerPixel, c3.maxBounceCount = 2, 4
    cases.append((c3, (216, 100, 240, 108) if mode == 1 else (216, 100, 264, 116)))
    for m, rect in cases:
        b = m.build_buffers()
        loop, cl = oracle.render_frame(*b, 1, rect, mode=mode)
        tree, ct = oracle.render_frame(*b, 1, rect, mode=mode, accel=True)
        _assert_bits(tree, loop, "search tree vs loop")
        assert (cl["rays"], cl["hits"], cl["sphereTests"]) == (ct["rays"], ct["hits"], ct["sphereTests"])


def test_oracle_search_tree_tie_break_and_degenerate_input(rtx, oracle):
    """Coincident triangles (equal dst: the first in buffer order wins, RayTracing.shader:287 strict <), zero-area and NaN
    triangles, and an empty scene go through the tree like through the loop."""
    m = rtx.scenes.mesh_test_scene(40, 24)
    p, s, t, mi = m.build_buffers()
    t = np.concatenate([t, t[:40], t[:40]]).copy()            # two more copies of the first 40 triangles, later in the buffer
    extra = np.zeros(2, mi.dtype)
    extra[:] = mi[0]
    extra["firstTriangleIndex"] = [len(t) - 80, len(t) - 40]
    extra["numTriangles"] = 40
    extra[1]["material"]["colour"] = (0.1, 0.9, 0.1, 1.0)     # visible if the tie went to the later copy
    mi2 = np.concatenate([mi, extra])
    t[-1]["posB"] = t[-1]["posA"]                             # zero area
    t[-2]["posC"] = (np.nan, 0, 0)
    loop, _ = oracle.render_frame(p, s, t, mi2, 0, mode=1)
    tree, _ = oracle.render_frame(p, s, t, mi2, 0, mode=1, accel=True)
    _assert_bits(tree, loop, "ties / degenerate")
    none, _ = oracle.render_frame(p, s[:0], t[:0], mi[:0], 0, accel=True)
    ref, _ = oracle.render_frame(p, s[:0], t[:0], mi[:0], 0)
    _assert_bits(none, ref, "empty scene")


def test_display_srgb8_known_values(oracle):
    """sRGB transfer: 0 -> 0, 1 -> 255, 0.0031308 -> 10 (linear segment), 0.5 -> 188, 0.2159 -> 128; >1 and NaN clamp."""
    px = np.array([[0.0, 1.0, 0.0031308, 1.0], [0.5, 0.2159, 2.0, 0.5], [np.nan, -1.0, 0.05, 1.0]], np.float32)
    out = oracle.display_srgb8(px[None])[0]
    assert out[0].tolist() == [0, 255, 10, 255]
    assert out[1].tolist() == [188, 128, 255, 128]
    assert out[2].tolist() == [0, 0, 63, 255]


def test_philox4x32_10_known_answers(oracle):
    """Random123 known-answer vectors for philox4x32-10 (Salmon et al., SC'11) — pins the counter-based RNG mode."""
    from ctypes import c_uint32
    L = oracle.lib
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        c, k, o = (c_uint32 * 4)(*ctr), (c_uint32 * 2)(*key), (c_uint32 * 4)()
        L.orc_philox4x32_10(c, k, o)
        assert tuple(o) == want


def test_philox_mode_is_a_different_stream_with_the_same_estimator(rtx, oracle):
    """rngMode = PHILOX: other noise than PCG, same expectation (means over many samples agree)."""
    m = rtx.scenes.config1(32, 32)
    m.numRaysPerPixel = 64
    b = list(m.build_buffers())
    pcg, _ = oracle.render_frame(*b, 0)
    p = b[0].copy(); p["rngMode"] = 1; b[0] = p
    phx, _ = oracle.render_frame(*b, 0)
    assert not np.array_equal(pcg, phx)
    lo = np.minimum(pcg[..., :3], 1).mean(), np.minimum(phx[..., :3], 1).mean()
    assert abs(lo[0] - lo[1]) < 0.05 * lo[0], lo
