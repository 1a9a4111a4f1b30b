"""GPU parity on the reference's own scenes (converted from Assets/Scenes/*.unity into tests/golden/scenes) and on the
instanced Chess workloads: against committed golden images, against the oracle on crops, and — at sizes the oracle
cannot reach — BVH kernel == flat-loop kernel on the GPU."""
import json
import os

import numpy as np
import pytest

from test_gpu_parity import assert_bitwise, bits_equal, run_gpu

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden_scene(rtx, name):
    z = np.load(os.path.join(GOLDEN, f"{name}_golden.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    from rtx_amd import unity_scene
    m = unity_scene.load_scene_npz(os.path.join(GOLDEN, "scenes", name + ".npz"), meta["width"], meta["height"])
    m.numRaysPerPixel, m.maxBounceCount = meta["rays"], meta["bounces"]
    return m, z, meta


@pytest.mark.parametrize("name", ["Chess", "Knight", "Reflective_Balls", "Balls_Outdoors"])
def test_reference_scene_matches_golden(rtx, tracer, name):
    m, z, meta = load_golden_scene(rtx, name)
    acc, last = run_gpu(tracer, m.build_buffers(), 0, meta["frames"])
    assert_bitwise(acc, z["accum"], f"{name} accumulated")
    assert_bitwise(last, z["last"], f"{name} last frame")


def test_config1_matches_golden(rtx, tracer):
    z = np.load(os.path.join(GOLDEN, "config1_golden.npz"))
    acc, last = run_gpu(tracer, rtx.scenes.config1().build_buffers(), 0, 1)
    assert_bitwise(last, z["frame0"], "config1 frame 0 vs committed golden")
    assert_bitwise(acc, z["accum0"], "config1 accum vs committed golden")


def test_chess_bvh_equals_flat_kernel_on_gpu(rtx, tracer):
    """Chess.unity with its own DOF settings at 480x270: the BVH kernel and the reference's literal loop (both on the
    GPU) give identical bits — a size the CPU oracle would need minutes for."""
    from rtx_amd import unity_scene
    m = unity_scene.load_scene_npz(os.path.join(GOLDEN, "scenes", "Chess.npz"), 480, 270)
    m.maxBounceCount = 6
    b = m.build_buffers()
    _, bvh = run_gpu(tracer, b, 3, 1)
    tracer.render_frame_flat(3)
    flat = tracer.read_last_frame()
    assert_bitwise(bvh, flat, "Chess BVH vs flat loop on GPU")


@pytest.mark.parametrize("mode", [0, 1])
def test_config3_crop_vs_oracle(rtx, oracle, tracer, mode):
    """~100k triangles: a 48x16 window of the 480x270 image, 2 rays, against the oracle (both intersect modes)."""
    m = rtx.scenes.config3(480, 270)
    m.numRaysPerPixel, m.maxBounceCount = 2, 4
    b = m.build_buffers()
    acc, last = run_gpu(tracer, b, 0, 1, mode=mode)
    rect = (216, 100, 264, 116)
    want, _ = oracle.render_frame(*b, 0, rect, mode=mode)
    assert_bitwise(last[rect[1]:rect[3], rect[0]:rect[2]], want, f"config3 crop mode {mode}")


def test_flat_vs_brute_difference_is_reported(rtx, tracer):
    """FLAT_CHUNKS (literal) and BRUTE (no chunk cull) may differ only where a chunk box rounds away a hit on its own
    face; count the pixels (informational, must stay tiny)."""
    m = rtx.scenes.config3(480, 270)
    m.numRaysPerPixel, m.maxBounceCount = 4, 4
    b = m.build_buffers()
    _, flat = run_gpu(tracer, b, 0, 1, mode=0)
    _, brute = run_gpu(tracer, b, 0, 1, mode=1)
    differing = int((~bits_equal(flat, brute)).any(-1).sum())
    print(f"FLAT_CHUNKS vs BRUTE differing pixels: {differing} of {flat.shape[0] * flat.shape[1]}")
    assert differing < 0.01 * flat.shape[0] * flat.shape[1]


def test_cpp_manager_renders_the_same_image(rtx, tracer, tmp_path):
    """C++ RayTracingManager::OnRenderImage (compiled host -> C-ABI) == the Python host's image, bit for bit."""
    from rtx_amd import unity_scene
    from rtx_amd.host_cpp_binding import CppScene
    mgr = rtx.scenes.mesh_test_scene(96, 64)
    path = str(tmp_path / "scene.unity")
    unity_scene.save_unity_scene(mgr, path)
    want, _ = run_gpu(tracer, mgr.build_buffers(), 0, 2)
    cpp = CppScene(path, 96, 64)
    got = cpp.render(2)
    got3 = cpp.render_multi(2, [0, 0, 0])           # the same manager through an rt_multi of three contexts on this GPU
    cpp.close()
    assert_bitwise(got, want, "C++ host")
    assert_bitwise(got3, want, "C++ host through rt_multi x3")


def test_cpp_manager_animates_through_the_device_geometry_pipeline(rtx, oracle, tmp_path):
    """The reference moves its meshes every frame (RayTracedMesh.cs:36-84).  The compiled manager, three pose changes in a row: the
    host path (world triangles re-marshalled and re-sent when they changed) == the on-device pipeline on one context (poses only) ==
    the same through an rt_multi of three contexts (rt_multi_upload_local_meshes / rt_multi_set_mesh_transforms) == the Python host's
    marshal of the final pose through the oracle."""
    from rtx_amd import unity_scene
    from rtx_amd.host_cpp_binding import CppScene
    h = rtx.host
    mgr = rtx.scenes.mesh_test_scene(88, 56)
    path = str(tmp_path / "scene.unity")
    unity_scene.save_unity_scene(mgr, path)
    cpp = CppScene(path, 88, 56)
    steps, shift = 3, np.float32([0.125, 0.0625, -0.25])
    q = np.float32([0.0, np.sin(0.2), 0.0, np.cos(0.2)])        # (float32 values handed to both hosts: no trigonometry on either side)
    host_path = cpp.render_animated(steps, 2, q, shift)
    dev_one = cpp.render_animated(steps, 2, q, shift, device_geometry=True)
    dev_multi = cpp.render_animated(steps, 2, q, shift, devices=[0, 0, 0], device_geometry=True)
    host_multi = cpp.render_animated(steps, 2, q, shift, devices=[0, 0])
    cpp.close()
    for _ in range(steps):                       # the same poses in the Python host (float32, UnityEngine's quaternion product)
        for i, mesh in enumerate(mgr.meshes):
            mesh.transform = h.Transform(position=mesh.transform.position + shift * np.float32(i + 1), rotation=h.quat_mul(q, mesh.transform.rotation),
                                         lossyScale=mesh.transform.lossyScale)
    want, _, _ = oracle.render(*mgr.build_buffers(), 0, 2)
    assert_bitwise(host_path, want, "C++ manager, animated, host path vs oracle")
    assert_bitwise(dev_one, want, "C++ manager, animated, device geometry")
    assert_bitwise(dev_multi, want, "C++ manager, animated, device geometry through rt_multi x3")
    assert_bitwise(host_multi, want, "C++ manager, animated, host path through rt_multi x2")


def test_config5_million_triangles_crop_vs_oracle(rtx, oracle, tracer):
    """configs[4]: 1,004,364 triangles, depth of field on — a 24x6 window of the 320x180 image against the oracle's flat
    loop over all 74k chunks (the deep BVH must return the reference's closest hit)."""
    m = rtx.scenes.config5(320, 180)
    m.numRaysPerPixel, m.maxBounceCount = 2, 3
    b = m.build_buffers()
    assert len(b[2]) == 1004364
    _, last = run_gpu(tracer, b, 0, 1)
    rect = (148, 60, 172, 66)
    want, cnt = oracle.render_frame(*b, 0, rect)
    assert cnt["boxTests"] > 1e7 and cnt["triTests"] > 1e4
    assert_bitwise(last[rect[1]:rect[3], rect[0]:rect[2]], want, "config5 crop")
    st = tracer.stats()
    assert st["numTriangles"] == 1004364 and st["numBvhNodes"] > 100000


def test_4k_bands_of_one_rank_match_the_full_frame(rtx, tracer):
    """configs[3] geometry at 3840x2160: the bands of rank 5 of 8 equal the same rows of the undivided frame."""
    m = rtx.scenes.config4()
    m.numRaysPerPixel, m.maxBounceCount = 1, 12
    b = m.build_buffers()
    full, _ = run_gpu(tracer, b, 0, 1)
    rows = rtx.distributed.band_rows(2160, 8, 5)
    tracer.set_bands(5, 8)
    tracer.reset_accum()
    tracer.render(0, 1)
    got = tracer.read_accum()
    tracer.set_rows(0, 2160)
    assert got.shape[0] == len(rows) == 272
    assert_bitwise(got, full[rows], "4K bands of rank 5/8")


def test_full_size_headline_frame_bvh_equals_flat_loop(rtx, tracer):
    """BASELINE's full size — 1920x1080, the 100,440-triangle scene, 8 bounces (2 rays per pixel to keep the flat loop's
    7,448 chunk tests per ray affordable): every pixel of the frame traced by k_stream, k_trace and the reference's
    literal chunk loop (all on the GPU) is identical; ray counts agree."""
    m = rtx.scenes.config3()
    m.numRaysPerPixel = 2
    b = m.build_buffers()
    _, stream = run_gpu(tracer, b, 0, 1, kernel=1)
    rays_stream = tracer.stats()["rays"]
    _, trace = run_gpu(tracer, b, 0, 1, kernel=0)
    rays_trace = tracer.stats()["rays"]
    tracer.render_frame_flat(0)
    flat = tracer.read_last_frame()
    rays_flat = tracer.stats()["rays"]
    assert stream.shape == (1080, 1920, 4)
    assert_bitwise(stream, flat, "k_stream vs flat loop, 1080p")
    assert_bitwise(trace, flat, "k_trace vs flat loop, 1080p")
    assert rays_stream == rays_trace == rays_flat > 4_000_000


# ---- full-size frames against the oracle (through the oracle's own search tree: tests/test_oracle_cpu.py shows tree == loop) ----

def _full_frame_vs_oracle(oracle, tracer, m, what, mode=0, frame=0, kernel=1):
    b = m.build_buffers()
    _, got = run_gpu(tracer, b, frame, 1, mode=mode, kernel=kernel)
    rays_gpu = tracer.stats()["rays"]
    want, cnt = oracle.render_frame(*b, frame, mode=mode, accel=True)
    assert_bitwise(got, want, what)
    return rays_gpu, cnt


def test_full_size_headline_frame_vs_oracle(rtx, oracle, tracer):
    """BASELINE's headline configuration exactly as benchmarked (1920x1080, 100,440 triangles, 64 rays per pixel, 8 bounces):
    every one of the 2,073,600 pixels of a frame equals the oracle's, and so does the number of rays traced (~2.4e8)."""
    m = rtx.scenes.config3()
    assert (m.numRaysPerPixel, m.maxBounceCount) == (64, 8)
    rays, cnt = _full_frame_vs_oracle(oracle, tracer, m, "config3 1080p x64 vs oracle", frame=5)
    assert rays == cnt["rays"] > 200_000_000


@pytest.mark.parametrize("name", ["Chess", "Knight", "Reflective_Balls", "Balls_Outdoors", "Suzanne", "Thumbnail"])
def test_reference_scene_full_hd_vs_oracle(rtx, oracle, tracer, name):
    """All six of the reference's own scenes at 1920x1080 with their serialized settings' bounce count, 1 ray per pixel, frame 3."""
    from rtx_amd import unity_scene
    m = unity_scene.load_scene_npz(os.path.join(GOLDEN, "scenes", name + ".npz"), 1920, 1080)
    m.numRaysPerPixel = 1
    _full_frame_vs_oracle(oracle, tracer, m, f"{name} 1080p vs oracle", frame=3, kernel=-1)


def test_million_triangle_frame_vs_oracle(rtx, oracle, tracer):
    """configs[4] (1,004,364 triangles, depth of field) at 960x540, 1 ray per pixel, whole frame."""
    m = rtx.scenes.config5(960, 540)
    m.numRaysPerPixel = 1
    _full_frame_vs_oracle(oracle, tracer, m, "config5 960x540 vs oracle")


def test_4k_frame_vs_oracle_brute_mode(rtx, oracle, tracer):
    """configs[3] (3840x2160, 12 bounces), 1 ray per pixel, BRUTE intersect mode (no chunk cull), whole frame."""
    m = rtx.scenes.config4()
    m.numRaysPerPixel = 1
    _full_frame_vs_oracle(oracle, tracer, m, "config4 4K vs oracle", mode=1)


def test_4k_frame_vs_oracle_flat_chunks_mode(rtx, oracle, tracer):
    """configs[3] (3840x2160, 12 bounces), 1 ray per pixel, the reference's literal FLAT_CHUNKS semantics (a triangle counts only if
    its chunk's RayBoundingBox passes), whole frame; the ray count is the oracle's."""
    m = rtx.scenes.config4()
    m.numRaysPerPixel = 1
    rays, cnt = _full_frame_vs_oracle(oracle, tracer, m, "config4 4K FLAT_CHUNKS vs oracle", mode=0, frame=2)
    assert rays == cnt["rays"] > 8_000_000


def test_config2_full_size_vs_oracle(rtx, oracle, tracer):
    """configs[1] exactly as BASELINE names it: 12 spheres, 1920x1080, 64 rays per pixel, 8 bounces — a whole frame (~3.4e8 rays)
    through the automatic kernel choice, every pixel and the ray count against the oracle's literal loop."""
    m = rtx.scenes.config2()
    assert (m.numRaysPerPixel, m.maxBounceCount, m.width, m.height) == (64, 8, 1920, 1080)
    b = m.build_buffers()
    _, got = run_gpu(tracer, b, 1, 1, kernel=-1)
    rays = tracer.stats()["rays"]
    want, cnt = oracle.render_frame(*b, 1)
    assert_bitwise(got, want, "config2 1080p x64 vs oracle")
    assert rays == cnt["rays"] > 300_000_000


def test_million_triangle_full_hd_vs_oracle(rtx, oracle, tracer):
    """configs[4] at its own resolution: 1,004,364 triangles, depth of field, 1920x1080, 1 ray per pixel, whole frame."""
    m = rtx.scenes.config5()
    m.numRaysPerPixel = 1
    rays, cnt = _full_frame_vs_oracle(oracle, tracer, m, "config5 1080p vs oracle", frame=1)
    assert rays == cnt["rays"]


def test_million_triangle_frame_as_benchmarked_vs_oracle(rtx, oracle, tracer):
    """configs[4] exactly as bench.py --config 5 runs it: 1,004,364 triangles, depth of field, 1920x1080, 64 rays per pixel, 8 bounces —
    every pixel of a whole frame and the ray count (~2.1e8) against the oracle."""
    m = rtx.scenes.config5()
    assert (m.numRaysPerPixel, m.maxBounceCount) == (64, 8)
    rays, cnt = _full_frame_vs_oracle(oracle, tracer, m, "config5 1080p x64 vs oracle", frame=2)
    assert rays == cnt["rays"] > 150_000_000


def test_4k_frame_as_benchmarked_vs_oracle(rtx, oracle, tracer):
    """configs[3] exactly as bench.py --config 4 runs it: 3840x2160, 64 rays per pixel, 12 bounces — all 8,294,400 pixels of a frame and
    the ray count (~9.7e8) against the oracle (its own search tree; the literal loop is covered at 1 ray per pixel above)."""
    m = rtx.scenes.config4()
    assert (m.numRaysPerPixel, m.maxBounceCount) == (64, 12)
    rays, cnt = _full_frame_vs_oracle(oracle, tracer, m, "config4 4K x64 vs oracle", frame=1)
    assert rays == cnt["rays"] > 800_000_000


def test_4k_sixty_four_frames_accumulated_vs_oracle(rtx, oracle, tracer):
    """configs[3]'s 4096 spp are 64 frames of 64 rays: here the 64 FRAMES at 3840x2160 and 12 bounces with 1 ray per pixel — four launches'
    worth of 16-frame groups through the saturating accumulate (Accumulate.shader:43-54), one rt_render call — against the oracle: all
    8,294,400 pixels of resultTexture and the ray count."""
    m = rtx.scenes.config4()
    m.numRaysPerPixel = 1
    b = m.build_buffers()
    acc, last = run_gpu(tracer, b, 0, 64, kernel=-1)
    st = tracer.stats()
    assert acc.shape == (2160, 3840, 4) and st["numRenderedFrames"] == 64
    want, want_last, cnt = oracle.render(*b, 0, 64, accel=True)
    assert_bitwise(last, want_last, "4K, frame 63")
    assert_bitwise(acc, want, "4K, 64 frames accumulated")
    assert st["rays"] == cnt["rays"] > 500_000_000


@pytest.mark.skipif(os.environ.get("RTX_FULL_TESTS", "0") != "1", reason="the whole 1024-spp job of configs[4] through the oracle: about five minutes of the "
                    "box's host cores; RTX_FULL_TESTS=1 runs it (its output of this round: profiles/validate_config5_r04.txt)")
def test_million_triangle_whole_job_vs_oracle(rtx, oracle, tracer):
    """configs[4] as a whole job: 1,004,364 triangles, depth of field, 1920x1080, 16 frames x 64 rays accumulated (1024 spp), 8 bounces —
    resultTexture and the ray count against the oracle (what tools/validate_headline.py does for the headline scene)."""
    b = rtx.scenes.config5().build_buffers()
    acc, _ = run_gpu(tracer, b, 0, 16, kernel=-1)
    rays = tracer.stats()["rays"]
    want, total = None, 0
    for f in range(16):
        cur, cnt = oracle.render_frame(*b, f, accel=True)
        if want is None:
            want = np.zeros_like(cur)
        oracle.accumulate(want, cur, f)
        total += cnt["rays"]
    assert_bitwise(acc, want, "config5 whole job, 16 frames accumulated")
    assert rays == total > 3_000_000_000
    print(f"config5 whole job: {rays} rays, resultTexture bit-identical to the oracle")


def test_headline_job_two_frames_accumulated_vs_oracle(rtx, oracle, tracer):
    """A two-frame cut of tools/validate_headline.py: frames 0 and 1 of the headline job (1920x1080, 100,440 triangles, 64 rays per
    pixel, 8 bounces) traced in ONE launch and accumulated — resultTexture and ray count equal the oracle's (~4.9e8 rays)."""
    b = rtx.scenes.config3().build_buffers()
    acc, _ = run_gpu(tracer, b, 0, 2, kernel=-1)
    rays = tracer.stats()["rays"]
    want = None
    total = 0
    for f in range(2):
        cur, cnt = oracle.render_frame(*b, f, accel=True)
        if want is None:
            want = np.zeros_like(cur)
        oracle.accumulate(want, cur, f)
        total += cnt["rays"]
    assert_bitwise(acc, want, "headline job, frames 0-1 accumulated")
    assert rays == total > 400_000_000


def test_sixteen_interleaved_frames_full_hd_vs_oracle(rtx, oracle, tracer):
    """The launch shape of the benchmark (one launch of 16 frames, a wave = 2x2 pixels x 16 frames) at 1920x1080 on the headline
    scene with 1 ray per pixel: the accumulated image of all 16 frames equals the oracle's."""
    m = rtx.scenes.config3()
    m.numRaysPerPixel = 1
    b = m.build_buffers()
    acc, _ = run_gpu(tracer, b, 0, 16, kernel=1)
    st = tracer.stats()
    assert st["lastFramesPerLaunch"] == 16 and st["lastFramesInterleaved"] == 16
    want, _, cnt = oracle.render(*b, 0, 16, accel=True)
    assert_bitwise(acc, want, "16 interleaved frames, 1080p")
    assert st["rays"] == cnt["rays"]


def test_render_scene_tool_writes_what_the_tracer_holds(rtx, tracer, tmp_path):
    """tools/render_scene.py (scene file -> OnRenderImage -> PNG / EXR): the EXR equals the accumulated resultTexture of the
    same render through the API, bit for bit; the PNG decodes to the display step's bytes."""
    import struct
    import subprocess
    import sys
    import zlib
    root = os.path.dirname(os.path.dirname(GOLDEN))
    scene = os.path.join(GOLDEN, "scenes", "Reflective_Balls.npz")
    png, exr = str(tmp_path / "o.png"), str(tmp_path / "o.exr")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "render_scene.py"), scene, "--width", "160", "--height", "90",
                          "--frames", "3", "--rays", "4", "--png", png, "--exr", exr], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "rays in" in out.stdout
    from rtx_amd import unity_scene
    m = unity_scene.load_scene_npz(scene, 160, 90, backend=tracer)
    m.numRaysPerPixel = 4
    tracer.set_option("kernel", -1)
    tracer.set_rows(0, 90)
    want = m.OnRenderImage(frames=3)
    got = rtx.imageio.read_exr(exr)
    assert_bitwise(got, want, "render_scene.py EXR vs API")
    disp = tracer.read_display()
    raw = open(png, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat = 8, b""
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        if tag == b"IDAT":
            idat += raw[pos + 8:pos + 8 + n]
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(90, 1 + 160 * 3)
    assert (rows[:, 0] == 0).all()
    assert (rows[:, 1:].reshape(90, 160, 3) == disp[::-1, :, :3]).all()


# ---- the counter-based mode at BASELINE's sizes (rt_params.rngMode = RT_RNG_PHILOX: per-sample counters, 16 sample lanes per pixel, the
# estimator's tree in the wave) against its oracle twin ----

def _philox(m):
    params, spheres, tris, infos = m.build_buffers()
    params = params.copy()
    params["rngMode"] = 1
    return params, spheres, tris, infos


def test_philox_headline_frame_vs_oracle(rtx, oracle, tracer):
    """configs[2] exactly as benchmarked with --rng philox (1920x1080, 100,440 triangles, 64 rays per pixel, 8 bounces): every pixel of a
    whole frame equals the oracle twin's, and so does the ray count (~2.4e8)."""
    b = _philox(rtx.scenes.config3())
    _, got = run_gpu(tracer, b, 4, 1, kernel=1)
    st = tracer.stats()
    want, cnt = oracle.render_frame(*b, 4, accel=True)
    assert st["lastSampleLanes"] == 16
    assert_bitwise(got, want, "philox, config3 1080p x64 vs oracle")
    assert st["rays"] == cnt["rays"] > 200_000_000


def test_philox_config2_full_size_vs_oracle(rtx, oracle, tracer):
    """configs[1] (12 spheres, 1920x1080, 64 rays per pixel, 8 bounces) in Philox mode: a whole frame (~3.4e8 rays) against the oracle twin."""
    b = _philox(rtx.scenes.config2())
    _, got = run_gpu(tracer, b, 1, 1, kernel=-1)
    st = tracer.stats()
    want, cnt = oracle.render_frame(*b, 1)
    assert st["lastKernel"] == 1 and st["lastSampleLanes"] == 16
    assert_bitwise(got, want, "philox, config2 1080p x64 vs oracle")
    assert st["rays"] == cnt["rays"] > 300_000_000


def test_philox_million_triangle_full_hd_vs_oracle(rtx, oracle, tracer):
    """configs[4] in Philox mode: 1,004,364 triangles, depth of field on (the defocus draws of block 0), 1920x1080, whole frame, 4 rays per
    pixel (4 sample lanes per pixel), two frames accumulated in one launch."""
    m = rtx.scenes.config5()
    m.numRaysPerPixel = 4
    b = _philox(m)
    acc, _ = run_gpu(tracer, b, 0, 2, kernel=1)
    st = tracer.stats()
    want, total = None, 0
    for f in range(2):
        cur, cnt = oracle.render_frame(*b, f, accel=True)
        if want is None:
            want = np.zeros_like(cur)
        oracle.accumulate(want, cur, f)
        total += cnt["rays"]
    assert st["lastSampleLanes"] == 4 and st["lastFramesPerLaunch"] == 2
    assert_bitwise(acc, want, "philox, config5 1080p x4, two frames")
    assert st["rays"] == total
