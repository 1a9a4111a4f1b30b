"""Multi-device path on the one GPU the test box has: N contexts on device 0 behind one rt_multi (the in-library row-band
decomposition + gather of include/rt.h), and two gloo ranks that each drive a real context."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_parity import assert_bitwise, run_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,size,frames", [(1, (64, 40), 3), (2, (104, 75), 5), (3, (93, 61), 4), (5, (40, 200), 16), (8, (64, 36), 2)])
def test_n_contexts_behind_rt_multi_equal_the_undivided_image(rtx, oracle, tracer, n, size, frames):
    """rt_multi with n contexts on device 0: interleaved bands, concurrent renders, one gather — bit-identical to one context
    rendering the whole image and to the oracle; rays add up.  (8 contexts on 36 rows: three of them own no band at all.)"""
    b = rtx.scenes.mesh_test_scene(*size).build_buffers()
    params, spheres, tris, infos = b
    with rtx.MultiTracer([0] * n) as mt:
        assert mt.count() == n
        mt.set_option("kernel", 1)
        mt.set_params(params)
        mt.upload(spheres=spheres, triangles=tris, meshinfo=infos)
        mt.render(1, frames)
        got = mt.read_accum()
        st = mt.stats()
        mt.reset_accum()
        mt.render(1, frames)                    # a second render over the reset state gives the same image
        again = mt.read_accum()
    ref, _ = run_gpu(tracer, b, 1, frames, kernel=1)
    want, _, cnt = oracle.render(*b, 1, frames)
    assert_bitwise(got, ref, f"rt_multi x{n} vs one context")
    assert_bitwise(got, want, f"rt_multi x{n} vs oracle")
    assert_bitwise(again, want, f"rt_multi x{n}, second render")
    assert st["rays"] == cnt["rays"]
    assert st["gatherMs"] >= 0.0


def test_rt_multi_in_philox_mode(rtx, oracle):
    """Three contexts behind one rt_multi in the counter-based mode (keys are global pixel indices, so the bands of every context draw what
    the undivided image draws): 16 rays per pixel = 16 sample lanes, two frames in one launch, against the oracle twin."""
    params, spheres, tris, infos = rtx.scenes.mesh_test_scene(90, 52).build_buffers()
    params = params.copy(); params["rngMode"] = 1; params["numRaysPerPixel"] = 16
    with rtx.MultiTracer([0] * 3) as mt:
        mt.set_params(params)
        mt.upload(spheres=spheres, triangles=tris, meshinfo=infos)
        mt.render(2, 2)
        got = mt.read_accum()
        st = mt.stats()
    want, _, cnt = oracle.render(params, spheres, tris, infos, 2, 2)
    assert_bitwise(got, want, "rt_multi x3, Philox mode")
    assert st["rays"] == cnt["rays"]


def test_rt_multi_through_the_peer_copy_api_on_one_device(rtx, oracle):
    """`peer_copies` = 1 sends the scene fan-out and the frame-end gather through hipMemcpyPeerAsync although every context sits on
    device 0: the calls, sizes and offsets of the branch a multi-GPU node takes run on the one-GPU box (the xGMI transfer itself does
    not).  Image and ray count stay the oracle's."""
    b = rtx.scenes.mesh_test_scene(88, 70).build_buffers()
    params, spheres, tris, infos = b
    with rtx.MultiTracer([0] * 4) as mt:
        mt.set_option("peer_copies", 1)
        mt.set_params(params)
        mt.upload(spheres=spheres, triangles=tris, meshinfo=infos)
        mt.render(0, 3)
        got = mt.read_accum()
        st, info = mt.stats(), mt.info()
    want, _, cnt = oracle.render(*b, 0, 3)
    assert_bitwise(got, want, "rt_multi x4, peer-copy API")
    assert st["rays"] == cnt["rays"] and info["bvhBuilds"] == 1


def test_rt_multi_builds_the_scene_once_whatever_the_number_of_contexts(rtx, oracle):
    """Eight contexts on device 0: the uploads go to the first context, which builds the scene (one BVH build); the other seven receive
    the built scene device to device and hold no host copy.  A second upload costs one more build; a camera that leaves the padded extent
    is handled per context without any build; images stay the oracle's."""
    b = rtx.scenes.mesh_test_scene(72, 64).build_buffers()
    params, spheres, tris, infos = b
    with rtx.MultiTracer([0] * 8) as mt:
        mt.set_params(params)
        mt.upload(spheres=spheres, triangles=tris, meshinfo=infos)
        mt.render(0, 2)
        got = mt.read_accum()
        info = mt.info()
        assert info["numContexts"] == 8 and info["bvhBuilds"] == 1, info
        assert info["device"] == [0] * 8 and info["peerAccess"] == [1] * 8 and info["lastSetupMs"] > 0.0
        want, _, _ = oracle.render(*b, 0, 2)
        assert_bitwise(got, want, "rt_multi x8, scene built once")
        mt.reset_accum(); mt.render(0, 2)
        assert mt.info()["bvhBuilds"] == 1                         # rendering again builds nothing
        # a builder option marks the scene for a rebuild: one more build, fanned out again
        mt.set_option("max_leaf", 4)
        mt.reset_accum(); mt.render(0, 2)
        assert mt.info()["bvhBuilds"] == 2
        assert_bitwise(mt.read_accum(), want, "rt_multi x8 after a rebuild with other leaves")
        mt.set_option("max_leaf", 2)
        # new content: fewer triangles (the first chunk only) — every context must see the new scene
        infos2 = infos[:1].copy()
        mt.upload(spheres=spheres, triangles=tris, meshinfo=infos2)
        mt.reset_accum(); mt.render(3, 1)
        want2, _, _ = oracle.render(params, spheres, tris, infos2, 3, 1)
        assert_bitwise(mt.read_accum(), want2, "rt_multi x8 after a second upload")
        assert mt.info()["bvhBuilds"] == 3


def test_rt_multi_error_paths(rtx):
    with pytest.raises(rtx.RtError):
        rtx.MultiTracer([99])
    with rtx.MultiTracer([0, 0]) as mt:
        with pytest.raises(rtx.RtError):
            mt.render(0, 1)                     # no params yet
        with pytest.raises(rtx.RtError):
            mt.set_option("no_such_option", 1)


def test_two_gloo_ranks_render_their_bands_on_the_gpu():
    """bench.py's N > 1 shape with real kernels: two processes, each with its own rt_ctx on GPU 0, bands + one gather (gloo)."""
    env = dict(os.environ, RTX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tests", "gloo_gpu_band_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "GLOO_GPU_BANDS_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_bench_started_plainly_with_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` started plainly: bench.py launches the two ranks itself (child torch.distributed.run) and relays one
    line.  Rehearsed on the box's one GPU: both ranks on device 0 (RTX_BENCH_DEVICE), gloo for the gather; real kernels, real bands."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTX_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--width", "320", "--height", "200",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["comm"]["world_seen"] == 2 and line["comm"]["backend"] == "gloo"
    assert line["value"] > 0 and all(r > 0 for r in line["per_rank"]["rays"]) and len(line["per_rank"]["rays"]) == 2


def test_bench_line_through_rccl_with_a_process_group_of_one():
    """bench.py's N > 1 code path on the one GPU of the box: RTX_BENCH_FORCE_DIST=1 makes a single rank initialise RCCL through
    torch.distributed (backend "nccl", device_id), run the barriers, the frame-end gather and the job report's all_gather for real.
    Not a scaling measurement — it shows that the calls the multi-GPU run makes are valid on this torch / RCCL build."""
    import json
    env = dict(os.environ, RTX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--width", "320", "--height", "200",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["comm"]["backend"] == "nccl" and line["comm"]["world_seen"] == 1 and line["n_gpus"] == 1
    assert line["value"] > 0 and line["per_rank"]["rays"][0] > 0 and "RCCL gather" in line["config"]["decomposition"]


@pytest.mark.parametrize("n,peer", [(1, 0), (3, 0), (8, 0), (3, 1)])
def test_moving_meshes_through_rt_multi(rtx, oracle, tracer, n, peer):
    """The reference moves meshes every frame (RayTracedMesh.cs:36-84, RayTracingManager.cs:135-164).  Through rt_multi: the local meshes go
    to every context once, then only poses (40 B per mesh) — every context transforms, builds and refits its own copy.  After each of two
    pose changes the assembled image == one context through the same pipeline == a fresh host-transformed upload == the oracle; the
    display step and a saved / restored accumulation state go through the handle as well."""
    h = rtx.host
    mgr = rtx.scenes.mesh_test_scene(88, 61)
    params, spheres, _, _ = mgr.build_buffers()
    with rtx.MultiTracer([0] * n) as mt:
        mt.set_option("kernel", 1); mt.set_option("peer_copies", peer)
        mt.set_params(params)
        mt.upload(spheres=spheres)
        mt.upload_local_meshes(*mgr.build_local_buffers(), len(mgr.meshes))
        for step in range(3):
            if step:
                for i, mesh in enumerate(mgr.meshes[1:]):
                    a = 0.3 * (i + 1) + 0.9 * step
                    q = h.quat_mul((0.0, np.sin(a / 2), 0.0, np.cos(a / 2)), mesh.transform.rotation)
                    mesh.transform = h.Transform(position=mesh.transform.position + np.float32([0.25 * step, 0.1 * i, -0.15]),
                                                 rotation=q, lossyScale=mesh.transform.lossyScale * np.float32(1.07))
            mt.set_mesh_transforms(mgr.build_transforms())
            mt.reset_accum()
            mt.render(2, 3)
            got = mt.read_accum()
            b = mgr.build_buffers()
            want, _, cnt = oracle.render(*b, 2, 3)
            assert_bitwise(got, want, f"rt_multi x{n} (peer_copies {peer}), moving meshes, step {step}: vs oracle")
            assert mt.stats()["rays"] == cnt["rays"]
            assert np.array_equal(mt.read_display(), oracle.display_srgb8(got))
        info = mt.info()
        assert info["bvhBuilds"] == n                     # one device build per context, then refits only
        # save / restore through the handle: 3 frames, unrelated work, restore, 2 more == 5 frames in one go
        saved = got.copy()
        mt.reset_accum(); mt.render(40, 1)
        mt.write_accum(saved, 5)                          # (frames 2..4 were rendered: the next frame index is 5)
        assert_bitwise(mt.read_accum(), saved, "restored state is visible before the next render")
        mt.render(5, 2)
        resumed = mt.read_accum()
        mt.reset_accum(); mt.render(2, 5)
        assert_bitwise(resumed, mt.read_accum(), f"rt_multi x{n}: resumed render vs uninterrupted")
        with pytest.raises(rtx.RtError):
            mt.write_accum(saved[:-1], 3)
    # one context through the same pipeline
    tracer.set_option("kernel", 1)
    tracer.set_rows(0, int(params["height"])); tracer.set_params(params); tracer.upload(spheres=spheres)
    tracer.upload_local_meshes(*mgr.build_local_buffers(), len(mgr.meshes)); tracer.set_mesh_transforms(mgr.build_transforms())
    tracer.reset_accum(); tracer.render(2, 3)
    assert_bitwise(got, tracer.read_accum(), f"rt_multi x{n} vs one context, device geometry")
