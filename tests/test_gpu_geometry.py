"""On-device geometry pipeline (rt_upload_local_meshes / rt_set_mesh_transforms): the GPU's world-space triangles and
chunk bounds equal the host marshal's bytes (RayTracedMesh.cs:56-94 restated in host.py), and images rendered through it —
including after a BVH *refit* for moved meshes — are bit-identical to uploading host-transformed buffers."""
import os

import numpy as np
import pytest

from test_gpu_parity import assert_bitwise, run_gpu

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def upload_local(tracer, mgr):
    params, spheres, _, _ = mgr.build_buffers()
    tracer.set_rows(0, int(params["height"]))
    tracer.set_params(params)
    tracer.upload(spheres=spheres)
    tracer.upload_local_meshes(*mgr.build_local_buffers(), len(mgr.meshes))
    tracer.set_mesh_transforms(mgr.build_transforms())
    return params


def bytes_equal(a, b):
    return a.tobytes() == b.tobytes()


@pytest.mark.parametrize("scene", ["mesh_test", "Knight", "config3"])
def test_device_world_geometry_equals_host_marshal(rtx, tracer, scene):
    from rtx_amd import unity_scene
    if scene == "mesh_test":
        mgr = rtx.scenes.mesh_test_scene(64, 48)
    elif scene == "Knight":
        mgr = unity_scene.load_scene_npz(os.path.join(GOLDEN, "scenes", "Knight.npz"), 64, 36)
    else:
        mgr = rtx.scenes.config3(64, 36)
    _, _, tris, infos = mgr.build_buffers()
    upload_local(tracer, mgr)
    dtris, dinfos = tracer.read_world_geometry()
    assert len(dtris) == len(tris) and len(dinfos) == len(infos)
    for k in tris.dtype.names:
        assert bytes_equal(dtris[k], tris[k]), f"{scene}: triangles.{k} differ"
    for k in ("firstTriangleIndex", "numTriangles", "boundsMin", "boundsMax"):
        assert bytes_equal(dinfos[k], infos[k]), f"{scene}: meshinfo.{k} differ"
    assert bytes_equal(dinfos["material"], infos["material"])
    print(f"{scene}: {len(tris)} triangles, device geometry pass {tracer.stats()['lastGeometryMs']:.3f} ms")


@pytest.mark.parametrize("mode", [0, 1])
def test_device_pipeline_image_equals_host_path(rtx, tracer, mode):
    mgr = rtx.scenes.mesh_test_scene(96, 64)
    mgr.intersectMode = mode
    want, want_last = run_gpu(tracer, mgr.build_buffers(), 1, 2, mode=mode)
    upload_local(tracer, mgr)
    tracer.reset_accum()
    tracer.render(1, 2)
    assert_bitwise(tracer.read_last_frame(), want_last, f"device geometry, mode {mode}, last frame")
    assert_bitwise(tracer.read_accum(), want, f"device geometry, mode {mode}, accum")


def test_refit_after_moving_meshes_equals_fresh_upload(rtx, tracer):
    """Animate: move / rotate / rescale the meshes -> only new transforms are sent, the BVH is refitted on the device."""
    mgr = rtx.scenes.mesh_test_scene(96, 64)
    upload_local(tracer, mgr)
    tracer.reset_accum()
    tracer.render(0, 1)
    nodes_before = tracer.stats()["numBvhNodes"]
    h = rtx.host
    for step in range(2):
        for i, mesh in enumerate(mgr.meshes[2:]):
            a = 0.4 * (i + 1) + 0.7 * step
            q = h.quat_mul((0.0, np.sin(a / 2), 0.0, np.cos(a / 2)), mesh.transform.rotation)
            mesh.transform = h.Transform(position=mesh.transform.position + np.float32([0.3 * step, 0.1 * i, -0.2]),
                                         rotation=q, lossyScale=mesh.transform.lossyScale * np.float32(1.1))
        tracer.set_mesh_transforms(mgr.build_transforms())          # 40 B per mesh
        tracer.reset_accum()
        tracer.render(3, 2)
        got, got_last = tracer.read_accum(), tracer.read_last_frame()
        assert tracer.stats()["numBvhNodes"] == nodes_before        # topology kept: refit, not rebuild
        dtris, dinfos = tracer.read_world_geometry()
        _, _, tris, infos = mgr.build_buffers()
        assert bytes_equal(dtris["posB"], tris["posB"]) and bytes_equal(dinfos["boundsMax"], infos["boundsMax"])
        want, want_last = run_gpu(tracer, mgr.build_buffers(), 3, 2)   # fresh host-transformed upload + BVH build
        assert_bitwise(got_last, want_last, f"refit step {step}, last frame")
        assert_bitwise(got, want, f"refit step {step}, accum")
        if step == 0:                                                # (run_gpu switched the context to the world path)
            upload_local(tracer, mgr)
            tracer.render(0, 1)
            nodes_before = tracer.stats()["numBvhNodes"]


def test_manager_device_geometry_flag(rtx, tracer):
    """RayTracingManager.OnRenderImage with deviceGeometry=True == the default host-marshal path."""
    mgr = rtx.scenes.mesh_test_scene(64, 48)
    want, _ = run_gpu(tracer, mgr.build_buffers(), 0, 2)
    mgr2 = rtx.scenes.mesh_test_scene(64, 48)
    mgr2.backend, mgr2.deviceGeometry = tracer, True
    mgr2.Start()
    got = mgr2.OnRenderImage(frames=2)
    assert_bitwise(got, want, "manager with deviceGeometry")


@pytest.mark.parametrize("seed", range(10))
def test_random_transforms_device_geometry_equals_host_marshal(rtx, tracer, seed):
    """Random poses per mesh — arbitrary unit quaternions (and a slightly non-unit one), non-uniform, negative and zero scales,
    translations up to 1e3 — through the device pipeline: world triangles and chunk bounds equal the host marshal's bytes,
    and the image equals the host path's (the second pose set goes through the BVH refit)."""
    rng = np.random.default_rng(500 + seed)
    mgr = rtx.scenes.mesh_test_scene(64, 48)

    def pose():
        for i, me in enumerate(mgr.meshes):
            q = rng.normal(size=4)
            q = q / np.linalg.norm(q) * (1.0001 if i == 1 else 1.0)
            s = 10.0 ** rng.uniform(-1.5, 0.8, 3) * rng.choice([1, 1, 1, -1], 3)
            if i == 2 and seed % 3 == 0:
                s[int(rng.integers(0, 3))] = 0.0
            big = 1e3 if seed % 4 == 3 else 6.0
            me.transform = rtx.host.Transform(position=tuple(rng.uniform(-big, big, 3)), rotation=tuple(q), lossyScale=tuple(s))

    pose()
    upload_local(tracer, mgr)
    for round_ in range(2):
        if round_ == 1:
            pose()
            tracer.set_mesh_transforms(mgr.build_transforms())      # same topology: refit
        _, _, tris, infos = mgr.build_buffers()
        tracer.reset_accum()
        tracer.render(round_, 1)
        got = tracer.read_last_frame()
        dtris, dinfos = tracer.read_world_geometry()
        for k in tris.dtype.names:
            assert bytes_equal(dtris[k], tris[k]), f"seed {seed} round {round_}: triangles.{k} differ"
        for k in ("boundsMin", "boundsMax"):
            assert bytes_equal(dinfos[k], infos[k]), f"seed {seed} round {round_}: meshinfo.{k} differ"
        _, want = run_gpu(tracer, mgr.build_buffers(), round_, 1, kernel=-1)
        assert_bitwise(got, want, f"seed {seed} round {round_}: image through the device pipeline")
        if round_ == 0:
            upload_local(tracer, mgr)                                # back to the local-mesh path for the refit round
