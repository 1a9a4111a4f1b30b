"""Edge cases the reference's semantics define: empty and ragged inputs, zero bounces, special material
flags, axis-aligned rays (exact zero direction components -> infinite reciprocals), degenerate primitives, depth of field.
Every case: GPU (both kernels) == oracle, bit for bit."""
import numpy as np
import pytest

from test_gpu_parity import assert_bitwise, run_gpu

pytestmark = pytest.mark.gpu


def check(rtx, oracle, tracer, mgr, frames=2, first=0, kernels=(0, 1), what=""):
    b = mgr.build_buffers()
    want, want_last, _ = oracle.render(*b, first, frames)
    for k in kernels:
        acc, last = run_gpu(tracer, b, first, frames, kernel=k)
        assert_bitwise(last, want_last, f"{what} kernel {k} last frame")
        assert_bitwise(acc, want, f"{what} kernel {k} accum")
    return want


def test_empty_scene_is_pure_environment(rtx, oracle, tracer):
    m = rtx.scenes.config1(48, 32)
    m.spheres = []
    img = check(rtx, oracle, tracer, m, what="empty scene")
    assert img[..., :3].max() > 0


def test_environment_disabled_and_nothing_to_hit_is_black(rtx, oracle, tracer):
    m = rtx.scenes.config1(32, 24)
    m.spheres = []
    m.environmentSettings.enabled = False
    img = check(rtx, oracle, tracer, m, what="black")
    assert np.all(img[..., :3] == 0) and np.all(img[..., 3] == 1)


def test_zero_bounces_one_ray(rtx, oracle, tracer):
    """maxBounceCount = 0: the loop bound is inclusive (RayTracing.shader:305), one cast per path."""
    m = rtx.scenes.mesh_test_scene(64, 40)
    m.maxBounceCount, m.numRaysPerPixel = 0, 1
    check(rtx, oracle, tracer, m, what="B=0")


def test_many_bounces_specular_hall_of_mirrors(rtx, oracle, tracer):
    m = rtx.scenes.config2(96, 54)
    m.maxBounceCount, m.numRaysPerPixel = 30, 3
    check(rtx, oracle, tracer, m, frames=1, what="B=30")


def test_ragged_chunks_zero_triangle_chunk_and_unreferenced_triangles(rtx, oracle, tracer):
    """A chunk with numTriangles = 0 and triangles no chunk refers to (the shader can never reach them)."""
    m = rtx.scenes.mesh_test_scene(64, 40)
    params, spheres, tris, infos = m.build_buffers()
    infos = infos.copy()
    empty = infos[:1].copy()
    empty["numTriangles"] = 0
    infos = np.concatenate([infos[:3], empty, infos[3:]])
    infos["numTriangles"][5] -= 3                      # the last 3 triangles of that chunk become unreachable
    b = (params, spheres, tris, infos)
    want, want_last, _ = oracle.render(*b, 0, 1)
    for k in (0, 1):
        acc, last = run_gpu(tracer, b, 0, 1, kernel=k)
        assert_bitwise(last, want_last, f"ragged chunks kernel {k}")


def test_axis_aligned_rays_zero_direction_components(rtx, oracle, tracer):
    """Identity camera, odd resolution, no jitter: the centre column / row have dir.x == 0 / dir.y == 0 exactly, so
    1/dir is infinite and the slab tests see inf and NaN (RayBoundingBox :179-186)."""
    h = rtx.host
    m = rtx.scenes.mesh_test_scene(33, 21)
    m.camera = h.Camera(h.Transform(position=(0.0, 1.0, -6.0)), fieldOfView=40.0, aspect=33 / 21)
    m.divergeStrength = m.defocusStrength = 0.0
    m.numRaysPerPixel, m.maxBounceCount = 2, 3
    b = m.build_buffers()
    assert float(b[0]["camLocalToWorld"][1]) == 0.0
    for mode in (0, 1):
        mm = m
        mm.intersectMode = mode
        check(rtx, oracle, tracer, mm, frames=1, what=f"axis-aligned mode {mode}")


def test_fixed_camera_origin_shortcut_and_the_cases_it_must_not_take(rtx, oracle, tracer):
    """defocusStrength = +-0 lets camera_ray take the camera position as the ray origin without the jitter arithmetic
    (pos + (+-0) = pos; rt_api.hip camera_origin_is_fixed).  A camera coordinate of -0 is the one value that arithmetic
    changes (-0 + +0 = +0), so such a camera has to stay on the general path; a tiny non-zero defocus as well."""
    m = rtx.scenes.mesh_test_scene(48, 30)
    m.numRaysPerPixel, m.maxBounceCount = 3, 3
    cases = {"+0 defocus": (0.0, None), "-0 defocus": (-0.0, None), "camera x = -0": (0.0, 0), "camera z = -0": (-0.0, 2),
             "denormal defocus": (1e-42, None)}
    for what, (defocus, neg_zero_axis) in cases.items():
        m.defocusStrength = defocus
        params, spheres, tris, infos = m.build_buffers()
        params = params.copy()
        params["defocusStrength"] = np.float32(defocus)
        if neg_zero_axis is not None:
            params["worldSpaceCameraPos"][neg_zero_axis] = np.float32(-0.0)
            params["camLocalToWorld"][4 * neg_zero_axis + 3] = np.float32(-0.0)
        b = (params, spheres, tris, infos)
        want, want_last, _ = oracle.render(*b, 0, 2)
        for k in (0, 1):
            acc, last = run_gpu(tracer, b, 0, 2, kernel=k)
            assert_bitwise(last, want_last, f"{what} kernel {k} last frame")
            assert_bitwise(acc, want, f"{what} kernel {k} accum")


def test_depth_of_field_and_invisible_light(rtx, oracle, tracer):
    """Chess.unity settings (defocus 180, diverge 1, focus 3.82) on the reference scene: exercises the defocus jitter
    (frag :377-378) and the InvisibleLightSource flag at bounce 0 (Trace :318-322)."""
    import os
    from rtx_amd import unity_scene
    m = unity_scene.load_scene_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes", "Chess.npz"), 80, 45)
    m.numRaysPerPixel, m.maxBounceCount = 2, 3
    assert m.defocusStrength == 180 and any(mesh.materials[0].flag == 2 for mesh in m.meshes)
    check(rtx, oracle, tracer, m, frames=1, kernels=(0, 1), what="DOF + invisible light")


def test_degenerate_primitives(rtx, oracle, tracer):
    """Zero-radius sphere, zero-area triangle, sliver below the absolute determinant threshold 1e-6 (RayTriangle :169)."""
    h = rtx.host
    m = rtx.scenes.mesh_test_scene(48, 32)
    m.spheres.append(h.RayTracedSphere(h.Transform(position=(0, 1, -2), lossyScale=(0, 0, 0)), h.RayTracingMaterial()))
    tri = np.zeros(3, rtx.TRIANGLE)
    tri["posA"], tri["posB"], tri["posC"] = [(0, 1, -3)] * 3, [(0, 1, -3), (1, 1, -3), (1e-4, 1, -3)], [(0, 1, -3), (2, 1, -3), (0, 1 + 1e-4, -3)]
    tri["normalA"] = tri["normalB"] = tri["normalC"] = (0, 0, -1)
    m.meshes.append(h.RayTracedMesh(h.Transform(), [h.RayTracingMaterial()], rtx.scenes.chunked(tri)))
    check(rtx, oracle, tracer, m, frames=1, what="degenerate")


def test_one_pixel_and_thin_images(rtx, oracle, tracer):
    for w, hgt in ((1, 1), (7, 1), (1, 9), (65, 3)):
        m = rtx.scenes.config1(w, hgt)
        check(rtx, oracle, tracer, m, frames=1, what=f"{w}x{hgt}")


def test_zero_sized_target_is_legal(rtx, tracer):
    params, spheres, tris, infos = rtx.scenes.config1(16, 8).build_buffers()
    params = params.copy()
    params["height"] = 0
    acc, last = run_gpu(tracer, (params, spheres, tris, infos), 0, 1)
    assert acc.shape == (0, 16, 4)


def test_error_codes_not_exceptions_across_the_abi(rtx, tracer):
    """Bad input comes back as a status code + message (the reference's only error is the 1500-triangle exception)."""
    params, spheres, tris, infos = rtx.scenes.mesh_test_scene(16, 16).build_buffers()
    bad = infos.copy()
    bad["firstTriangleIndex"][-1] = len(tris)           # range beyond the triangle buffer
    tracer.set_rows(0, 16)
    tracer.set_params(params)
    tracer.upload(spheres=spheres, triangles=tris, meshinfo=bad)
    with pytest.raises(rtx.RtError, match="beyond"):
        tracer.render(0, 1)
    overlap = infos.copy()
    overlap["firstTriangleIndex"][1] = overlap["firstTriangleIndex"][0]
    tracer.upload(meshinfo=overlap)
    with pytest.raises(rtx.RtError, match="referenced by chunks"):
        tracer.render(0, 1)
    with pytest.raises(rtx.RtError, match="unknown option"):
        tracer.set_option("no_such_option", 1)
    p = params.copy()
    p["rngMode"] = 7
    with pytest.raises(rtx.RtError, match="rngMode"):
        tracer.set_params(p)
    tracer.upload(meshinfo=infos)                       # leave the shared context usable
    tracer.set_params(params)
    tracer.render(0, 1)


def test_zero_normals_make_nan_rays_that_hit_nothing(rtx, oracle, tracer):
    """A zero vertex normal makes normalize() return NaN (RayTracing.shader:293): the bounce ray is all NaN, can hit nothing,
    and ends in the environment (whose saturate() turns NaN into 0).  Its slab tests are NaN on every axis, so a BVH must not let it "enter" the empty
    child slots of small meshes (found by tests/test_gpu_fuzz.py: k_stream decoded an empty slot as a leaf)."""
    m = rtx.scenes.mesh_test_scene(64, 40)
    m.maxBounceCount, m.numRaysPerPixel = 5, 2
    b = list(m.build_buffers())
    tris = b[2].copy()
    tris["normalA"][::2] = 0; tris["normalB"][::2] = 0; tris["normalC"][::2] = 0
    b[2] = tris
    want, want_last, _ = oracle.render(*b, 1, 2)
    for k in (0, 1):
        acc, last = run_gpu(tracer, tuple(b), 1, 2, kernel=k)
        assert_bitwise(last, want_last, f"NaN rays, kernel {k}, last frame")
        assert_bitwise(acc, want, f"NaN rays, kernel {k}, accum")


def test_resume_from_a_saved_accumulation_state(rtx, oracle, tracer):
    """rt_write_accum: resultTexture + frame counter are the whole state the reference carries between frames
    (RayTracingManager.cs:26,33); 3 frames, save, other work, restore, 4 more == 7 frames in one go == the oracle."""
    b = rtx.scenes.config1(80, 48).build_buffers()
    full, _ = run_gpu(tracer, b, 0, 7)
    part, _ = run_gpu(tracer, b, 0, 3)
    assert tracer.stats()["numRenderedFrames"] == 3
    saved = part.copy()
    tracer.reset_accum()
    tracer.render(11, 2)                                  # unrelated frames in between
    tracer.write_accum(saved, 3)
    assert tracer.stats()["numRenderedFrames"] == 3
    tracer.render(3, 4)
    got = tracer.read_accum()
    want, _, _ = oracle.render(*b, 0, 7)
    assert_bitwise(got, full, "resumed render vs uninterrupted")
    assert_bitwise(got, want, "resumed render vs oracle")
    with pytest.raises(rtx.RtError):
        tracer.write_accum(saved[:-1], 3)


def test_nan_and_zero_direction_rays_do_not_walk_the_tree(rtx, oracle, tracer):
    """Rays with a NaN in them (zero normals -> normalize(0) = NaN) or a zero direction complete without traversal in every
    kernel: node visits per ray stay at the level of the traceable rays, and the images are the oracle's."""
    m = rtx.scenes.mesh_test_scene(48, 32)
    params, spheres, tris, infos = m.build_buffers()
    tris = tris.copy()
    for f in ("normalA", "normalB", "normalC"):
        tris[f] = 0.0                                     # every triangle hit makes a NaN bounce ray
    b = (params, spheres, tris, infos)
    want, want_last, cnt = oracle.render(*b, 0, 2)
    for kernel in (0, 1):
        tracer.set_option("kernel", kernel)
        tracer.set_params(params); tracer.upload(spheres=spheres, triangles=tris, meshinfo=infos)
        tracer.set_rows(0, int(params["height"]))
        tracer.reset_accum()
        tracer.render_counting(0, 2)
        st = tracer.stats()
        assert_bitwise(tracer.read_accum(), want, f"kernel {kernel}: NaN-ray scene")
        assert st["rays"] == cnt["rays"]
        nn = st["numBvhNodes"]
        # a NaN ray that walked the tree would visit all nn nodes; traceable rays visit a few dozen at most
        assert st["nodeVisits"] < st["rays"] * min(nn, 60), (kernel, st["nodeVisits"], st["rays"], nn)
    tracer.set_option("kernel", 0)


def test_automatic_kernel_choice_without_the_costliest_first_order(rtx, oracle, tracer):
    """tile_lpt = 0: there is no tile order to wait for; the automatic choice still measures both kernels on the first frames,
    decides, and batches the rest (it used to stay on single-frame k_trace launches for ever)."""
    b = rtx.scenes.mesh_test_scene(96, 64).build_buffers()
    tracer.set_option("tile_lpt", 0)
    try:
        acc, last = run_gpu(tracer, b, 0, 9, kernel=-1)
        st = tracer.stats()
        assert st["autoKernel"] in (0, 1)
        tracer.render(9, 8)
        assert tracer.stats()["lastFramesPerLaunch"] == 8
        acc = tracer.read_accum()
    finally:
        tracer.set_option("tile_lpt", 1)
    want, _, _ = oracle.render(*b, 0, 17)
    assert_bitwise(acc, want, "auto kernel, tile_lpt = 0")


def test_camera_drifting_away_from_the_geometry_widens_the_padding_without_a_rebuild(rtx, oracle, tracer):
    """World-space uploads: the BVH boxes are padded for ray origins up to a magnitude G (camera, spheres, triangles).  A camera
    beyond the triangles' extent that keeps moving outward used to rebuild the whole scene on the host every frame; now the
    padding has headroom (2x) and is widened on the device (refit + f16 nodes) once per doubling: no BVH build, image == oracle."""
    h = rtx.host
    m = rtx.scenes.mesh_test_scene(64, 40)
    m.numRaysPerPixel, m.maxBounceCount = 2, 3
    b = m.build_buffers()
    params, spheres, tris, infos = b
    extent = float(np.abs(np.concatenate([tris["posA"], tris["posB"], tris["posC"]])).max()) if len(tris) else 1.0
    for k in (0, 1):
        tracer.set_option("kernel", k)
        tracer.set_rows(0, int(params["height"]))
        tracer.upload(spheres=spheres[:0], triangles=tris, meshinfo=infos)
        builds0 = repads0 = None
        dist = 1.5 * extent
        for step in range(7):                                  # 1.5x, 1.95x, ... 7.2x the extent: a few doublings
            p = params.copy()
            pos = np.float32([0.3, 1.0 + 0.1 * step, -dist])
            p["worldSpaceCameraPos"] = pos
            mtx = p["camLocalToWorld"].copy(); mtx[3], mtx[7], mtx[11] = pos; p["camLocalToWorld"] = mtx
            tracer.set_params(p)
            tracer.reset_accum()
            tracer.render_frame(step)
            got = tracer.read_last_frame()
            st = tracer.stats()
            if builds0 is None:
                builds0, repads0 = st["bvhBuilds"], st["bvhRepads"]
            want, _ = oracle.render_frame(p, spheres[:0], tris, infos, step)
            assert_bitwise(got, want, f"kernel {k}, camera at {dist / extent:.2f}x the extent")
            assert st["bvhBuilds"] == builds0, f"camera drift triggered a BVH build at step {step}"
            dist *= 1.3
        assert 1 <= st["bvhRepads"] - repads0 <= 3, st["bvhRepads"] - repads0
    tracer.set_option("kernel", 0)


def test_philox_mode_at_the_largest_accepted_bounce_count(rtx, oracle, tracer):
    """A closed room of white mirrors keeps every path alive (Russian roulette's p = 1: Trace :338-342), so only the loop bound (:305)
    ends it.  The Philox instantiation keeps sample and bounce in one register (bounce in the high half); the host accepts up to 32000
    bounces so that the packed counter never reaches the sign bit.  8x8 pixels x 1 ray x 32001 casts: image and ray count == oracle twin;
    one bounce more is refused."""
    m = rtx.scenes.config2(8, 8)
    for s in m.spheres:
        s.material.colour = (1, 1, 1, 1); s.material.specularColour = (1, 1, 1, 1); s.material.emissionStrength = 0.0
        s.material.specularProbability = 1.0; s.material.smoothness = 1.0; s.material.flag = 0
    m.numRaysPerPixel, m.maxBounceCount = 1, 32000
    params, spheres, tris, infos = m.build_buffers()
    params = params.copy(); params["rngMode"] = 1
    b = (params, spheres, tris, infos)
    acc, last = run_gpu(tracer, b, 0, 1, kernel=1)
    st = tracer.stats()
    want, want_last, cnt = oracle.render(*b, 0, 1)
    assert cnt["rays"] > 60 * 32001                       # nearly every path ran into the bound
    assert st["rays"] == cnt["rays"]
    assert_bitwise(acc, want, "philox, 32000 bounces")
    params["maxBounceCount"] = 32001
    tracer.set_params(params)
    with pytest.raises(rtx.RtError):
        tracer.render(0, 1)
    params["maxBounceCount"] = 3
    tracer.set_params(params)
