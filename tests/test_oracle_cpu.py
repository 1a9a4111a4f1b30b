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


def _philox4x32_10(ctr, key):
    """Philox4x32-10 in plain Python integers (Salmon et al., SC'11) — independent of the oracle's C."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def test_philox_python_twin_matches_the_known_answers():
    assert _philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert _philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


@pytest.mark.parametrize("rays", [1, 3, 5, 16, 37, 64])
def test_philox_mode_addressing_and_estimator_tree_against_a_python_restatement(rtx, oracle, rays):
    """The Philox mode's definition in a second, independent form (DESIGN.md "Counter-based mode"): on a scene with nothing to
    hit, a sample's radiance is the environment light along its camera ray, which depends only on block 0 of
    philox(key = (pixelIndex, Frame), counter = (0, sample, 0, 0)).  This restates frag :356-389 for that case in numpy float32
    (transcendentals through the oracle's exported om_* functions), sums the samples in the defined tree — sub-stream s mod S added in
    order, then pairs (k, k+1), (k, k+2), ... — and expects the oracle's image bit for bit: key / counter layout, draw order and
    tree order are all pinned by it."""
    f32 = np.float32
    m = rtx.scenes.config1(7, 5)
    m.spheres = []
    m.numRaysPerPixel = rays
    m.defocusStrength, m.divergeStrength = 2.5, 1.75
    params, spheres, tris, infos = m.build_buffers()
    params = params.copy(); params["rngMode"] = 1
    frame = 11
    got, _ = oracle.render_frame(params, spheres, tris, infos, frame)
    L = oracle.lib
    W, H = int(params["width"]), int(params["height"])
    M = params["camLocalToWorld"].astype(f32)
    vp = params["viewParams"].astype(f32)
    right, up, pos = (M[0], M[4], M[8]), (M[1], M[5], M[9]), tuple(params["worldSpaceCameraPos"].astype(f32))
    defocus, diverge = f32(params["defocusStrength"]), f32(params["divergeStrength"])
    assert bool(params["environmentEnabled"])
    ground, horizon, zenith = (params[k].astype(f32) for k in ("groundColour", "skyColourHorizon", "skyColourZenith"))
    light_dir = params["worldSpaceLightPos0"].astype(f32)
    sun_focus, sun_intensity = f32(params["sunFocus"]), f32(params["sunIntensity"])
    S = 16 if rays >= 16 else 4 if rays >= 4 else 1
    assert L.orc_philox_substreams(rays) == S

    def sat(x):
        return f32(min(max(x, f32(0)), f32(1)))

    def smoothstep(e0, e1, x):
        t = sat(f32(f32(x - e0) / f32(e1 - e0)))
        return f32(f32(t * t) * f32(f32(3) - f32(f32(2) * t)))

    def lerp(a, b, t):
        return f32(a + f32(t * f32(b - a)))

    def u01(r):
        return f32(f32(np.uint32(r)) * f32(2.3283064365386963e-10))

    def point_in_circle(ua, ub):
        angle = f32(f32(ua * f32(2.0)) * f32(3.1415))
        s = f32(np.sqrt(ub))
        return f32(f32(L.om_cos(angle)) * s), f32(f32(L.om_sin(angle)) * s)

    def env(d):
        sky_t = f32(L.om_pow(smoothstep(f32(0), f32(0.4), d[1]), f32(0.35)))
        g2s = smoothstep(f32(-0.01), f32(0), d[1])
        sky = [lerp(horizon[c], zenith[c], sky_t) for c in range(3)]
        dot = f32(f32(f32(d[0] * light_dir[0]) + f32(d[1] * light_dir[1])) + f32(d[2] * light_dir[2]))
        sun = f32(f32(L.om_pow(f32(max(f32(0), dot)), sun_focus)) * sun_intensity)
        sun_term = f32(sun * (f32(1) if g2s >= 1 else f32(0)))
        return [f32(lerp(ground[c], sky[c], g2s) + sun_term) for c in range(3)]

    want = np.zeros((H, W, 4), f32)
    with np.errstate(all="ignore"):
        for y in range(H):
            for x in range(W):
                uvx, uvy = f32(f32(f32(x) + f32(0.5)) / f32(W)), f32(f32(f32(y) + f32(0.5)) / f32(H))
                lx, ly, lz = f32(f32(uvx - f32(0.5)) * vp[0]), f32(f32(uvy - f32(0.5)) * vp[1]), f32(f32(1) * vp[2])
                focus = [f32(f32(f32(f32(M[4 * r] * lx) + f32(M[4 * r + 1] * ly)) + f32(M[4 * r + 2] * lz)) + f32(M[4 * r + 3] * f32(1))) for r in range(3)]
                part = [[f32(0)] * 3 for _ in range(S)]
                for s in range(rays):
                    w = _philox4x32_10((0, s, 0, 0), (y * W + x, frame))
                    jx, jy = point_in_circle(u01(w[0]), u01(w[1]))
                    jx, jy = f32(f32(jx * defocus) / f32(W)), f32(f32(jy * defocus) / f32(W))
                    origin = [f32(f32(pos[c] + f32(right[c] * jx)) + f32(up[c] * jy)) for c in range(3)]
                    jx, jy = point_in_circle(u01(w[2]), u01(w[3]))
                    jx, jy = f32(f32(jx * diverge) / f32(W)), f32(f32(jy * diverge) / f32(W))
                    target = [f32(f32(focus[c] + f32(right[c] * jx)) + f32(up[c] * jy)) for c in range(3)]
                    v = [f32(target[c] - origin[c]) for c in range(3)]
                    ln = f32(np.sqrt(f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))))
                    d = [f32(v[c] / ln) for c in range(3)]
                    e = env(d)
                    light = [f32(f32(0) + f32(e[c] * f32(1))) for c in range(3)]            # incomingLight += env * rayColour (:346)
                    part[s % S] = [f32(part[s % S][c] + light[c]) for c in range(3)]
                step = 1
                while step < S:
                    for k in range(0, S, 2 * step):
                        part[k] = [f32(part[k][c] + part[k + step][c]) for c in range(3)]
                    step *= 2
                want[y, x] = [f32(part[0][c] / f32(rays)) for c in range(3)] + [f32(1)]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
