"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle, bit for bit, on seeded inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def assert_bitwise(got, want, what):
    same = bits_equal(got, want)
    if not same.all():
        bad = np.argwhere(~same.all(-1))
        y, x = bad[0]
        raise AssertionError(f"{what}: {len(bad)} of {got.shape[0] * got.shape[1]} pixels differ; first at (x={x}, y={y}): "
                             f"gpu {got[y, x]} oracle {want[y, x]}")


def run_gpu(tracer, buffers, first, n, mode=None, rows=None, kernel=0, shade_threshold=48, tile_sync=1):
    params, spheres, tris, infos = buffers
    p = params.copy()
    if mode is not None:
        p["intersectMode"] = mode
    H = int(p["height"])
    tracer.set_rows(*(rows if rows else (0, H)))
    tracer.set_option("kernel", kernel)
    tracer.set_option("shade_threshold", shade_threshold)
    tracer.set_option("tile_sync", tile_sync)
    tracer.set_params(p)
    tracer.upload(spheres=spheres, triangles=tris, meshinfo=infos)
    tracer.reset_accum()
    tracer.render(first, n)
    return tracer.read_accum(), tracer.read_last_frame()


@pytest.mark.parametrize("kernel", [0, 1])
def test_config1_spheres_bitwise(rtx, oracle, tracer, kernel):
    """configs[0]: 16 spheres, 256x256, 4 rays, 3 bounces — full image, frame 0."""
    b = rtx.scenes.config1().build_buffers()
    acc, last = run_gpu(tracer, b, 0, 1, kernel=kernel)
    want_acc, want_last, _ = oracle.render(*b, 0, 1)
    assert_bitwise(last, want_last, "config1 currentFrame")
    assert_bitwise(acc, want_acc, "config1 resultTexture")


def test_config1_accumulate_three_frames(rtx, oracle, tracer):
    """Frames 0..2 accumulated in order (sun at 200x clamps: the running average is not a plain mean)."""
    b = rtx.scenes.config1(96, 96).build_buffers()
    acc, _ = run_gpu(tracer, b, 0, 3)
    want_acc, _, _ = oracle.render(*b, 0, 3)
    assert_bitwise(acc, want_acc, "config1 3-frame accumulation")
    assert want_acc.max() <= 1.0


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("mode", [0, 1])
def test_mesh_scene_bvh_bitwise(rtx, oracle, tracer, mode, kernel):
    """Triangles through the BVH == the reference's flat chunk loop (mode 0) / brute force (mode 1), for both
    megakernel schedules."""
    b = rtx.scenes.mesh_test_scene(96, 64).build_buffers()
    acc, last = run_gpu(tracer, b, 5, 2, mode=mode, kernel=kernel)
    want_acc, want_last, _ = oracle.render(*b, 5, 2, mode=mode)
    assert_bitwise(last, want_last, f"mesh scene frame 6 mode {mode}")
    assert_bitwise(acc, want_acc, f"mesh scene accum mode {mode}")


def test_flat_gpu_kernel_matches_oracle(rtx, oracle, tracer):
    """The validation twin (the reference's literal loop on the GPU) is itself bit-identical to the oracle."""
    b = rtx.scenes.mesh_test_scene(64, 48).build_buffers()
    params, spheres, tris, infos = b
    tracer.set_rows(0, int(params["height"]))
    tracer.set_params(params)
    tracer.upload(spheres=spheres, triangles=tris, meshinfo=infos)
    tracer.reset_accum()
    tracer.render_frame_flat(0)
    got = tracer.read_last_frame()
    want, _ = oracle.render_frame(*b, 0)
    assert_bitwise(got, want, "flat GPU kernel")


def test_row_strip_is_decomposition_invariant(rtx, oracle, tracer):
    """Rows [20,45) rendered alone equal the same rows of the full image (global pixel seeds)."""
    b = rtx.scenes.mesh_test_scene(80, 64).build_buffers()
    full, _ = run_gpu(tracer, b, 0, 2)
    strip, _ = run_gpu(tracer, b, 0, 2, rows=(20, 25))
    assert strip.shape[0] == 25
    assert_bitwise(strip, full[20:45], "row strip")


def test_counting_build_matches_oracle_ray_count(rtx, oracle, tracer):
    b = rtx.scenes.config1(64, 64).build_buffers()
    params, spheres, tris, infos = b
    tracer.set_rows(0, 64)
    tracer.set_params(params)
    tracer.upload(spheres=spheres, triangles=tris, meshinfo=infos)
    tracer.reset_accum()
    tracer.render_counting(0, 1)
    st = tracer.stats()
    _, cnt = oracle.render_frame(*b, 0)
    assert st["rays"] == cnt["rays"] and st["hits"] == cnt["hits"] and st["sphereTests"] == cnt["sphereTests"]


@pytest.mark.parametrize("node_min", [1, 24, 64])
def test_node_loop_cap_does_not_change_the_image(rtx, oracle, tracer, node_min):
    """k_stream's node loop may hand over to the leaves while some lanes still hold internal nodes (option node_min):
    those lanes must simply wait — image and work counters stay the oracle's."""
    b = rtx.scenes.mesh_test_scene(93, 61).build_buffers()
    tracer.set_option("node_min", node_min)
    try:
        got, got_last = run_gpu(tracer, b, 2, 2, kernel=1)
        tracer.reset_accum()
        tracer.render_counting(2, 2)
        st = tracer.stats()
    finally:
        tracer.set_option("node_min", 10)
    want, want_last, cnt = oracle.render(*b, 2, 2)
    assert_bitwise(got_last, want_last, f"node_min={node_min}: last frame")
    assert_bitwise(got, want, f"node_min={node_min}: accum")
    assert st["rays"] == cnt["rays"] and st["hits"] == cnt["hits"]


@pytest.mark.parametrize("tile_sync", [0, 1])
@pytest.mark.parametrize("threshold", [1, 13, 64])
def test_schedule_knobs_do_not_change_the_image(rtx, tracer, threshold, tile_sync):
    """Resumable-traversal kernel at extreme shade thresholds, tile-at-a-time and pixel-at-a-time refill == the
    tile-per-wave kernel, odd image size (partial tiles)."""
    b = rtx.scenes.mesh_test_scene(93, 61).build_buffers()
    ref, ref_last = run_gpu(tracer, b, 2, 2, kernel=0)
    got, got_last = run_gpu(tracer, b, 2, 2, kernel=1, shade_threshold=threshold, tile_sync=tile_sync)
    assert_bitwise(got, ref, f"stream(threshold={threshold}) vs tile kernel, accum")
    assert_bitwise(got_last, ref_last, f"stream(threshold={threshold}) vs tile kernel, last frame")


@pytest.mark.parametrize("stream_stack", [4, 7, 12])
def test_stream_stack_spill_does_not_change_the_image(rtx, tracer, stream_stack):
    """k_stream with only a few traversal-stack entries per lane in LDS (the rest spills to global memory) == tile kernel:
    image and ray count; the mesh scene's BVH needs more entries than any of the three caps."""
    b = rtx.scenes.mesh_test_scene(93, 61).build_buffers()
    ref, ref_last = run_gpu(tracer, b, 2, 2, kernel=0)
    rays_ref = tracer.stats()["rays"]
    assert tracer.stats()["bvhMaxStack"] + 3 > 12
    tracer.set_option("stream_stack", stream_stack)
    try:
        got, got_last = run_gpu(tracer, b, 2, 2, kernel=1)
        rays = tracer.stats()["rays"]
    finally:
        tracer.set_option("stream_stack", 0)
    assert_bitwise(got_last, ref_last, f"stream(stack={stream_stack}) vs tile kernel, last frame")
    assert_bitwise(got, ref, f"stream(stack={stream_stack}) vs tile kernel, accum")
    assert rays == rays_ref


@pytest.mark.parametrize("kernel", [0, 1])
def test_interleaved_bands_are_decomposition_invariant(rtx, tracer, kernel):
    """8-row bands dealt round-robin to 3 'ranks' reassemble to the undivided image (61 rows: partial last band)."""
    b = rtx.scenes.mesh_test_scene(80, 61).build_buffers()
    full, _ = run_gpu(tracer, b, 0, 2, kernel=kernel)
    out = np.zeros_like(full)
    for rank in range(3):
        rows = rtx.distributed.band_rows(61, 3, rank)
        params, spheres, tris, infos = b
        tracer.set_params(params)
        tracer.set_bands(rank, 3)
        tracer.reset_accum()
        tracer.render(0, 2)
        got = tracer.read_accum()
        assert got.shape[0] == len(rows)
        out[rows] = got
    tracer.set_rows(0, 61)
    assert_bitwise(out, full, "interleaved bands")


def test_display_srgb8_matches_oracle(rtx, oracle, tracer, tmp_path):
    """Display step after the path (linear -> sRGB8, RayTracingManager.cs:84): GPU kernel == oracle on every byte."""
    b = rtx.scenes.config1(160, 96).build_buffers()
    acc, _ = run_gpu(tracer, b, 0, 2)
    got = tracer.read_display()
    want = oracle.display_srgb8(acc)
    assert got.shape == (96, 160, 4) and got.dtype == np.uint8
    assert np.array_equal(got, want), f"{int((got != want).any(-1).sum())} pixels differ"
    assert got[..., 3].min() == 255 and got[..., :3].max() == 255 and got[..., :3].min() < 40
    rtx.imageio.write_png(str(tmp_path / "c1.png"), got)
    rtx.imageio.write_pfm(str(tmp_path / "c1.pfm"), acc)
    assert (tmp_path / "c1.png").stat().st_size > 1000


def _philox(buffers, rays=None, **extra):
    params, spheres, tris, infos = buffers
    params = params.copy()
    params["rngMode"] = 1
    if rays is not None:
        params["numRaysPerPixel"] = rays
    for k, v in extra.items():
        params[k] = v
    return params, spheres, tris, infos


@pytest.mark.parametrize("scene", ["spheres", "mesh"])
@pytest.mark.parametrize("rays", [1, 3, 5, 16, 21, 64])
def test_philox_mode_bitwise_vs_oracle(rtx, oracle, tracer, scene, rays):
    """rt_params.rngMode = RT_RNG_PHILOX: Philox4x32-10, key (pixelIndex, Frame), counter (block, sample); a pixel's samples sit on
    1 / 4 / 16 lanes of a wave (NumRaysPerPixel < 4 / < 16 / >= 16), each lane sums its sub-stream, the wave adds the sub-streams in
    the estimator's fixed tree.  GPU == oracle twin, bit for bit, for every sub-stream shape and ragged last rounds (21 = 16 + 5)."""
    m = rtx.scenes.config1(61, 43) if scene == "spheres" else rtx.scenes.mesh_test_scene(61, 43)
    b = _philox(m.build_buffers(), rays)
    want, want_last, cnt = oracle.render(*b, 3, 2)
    acc, last = run_gpu(tracer, b, 3, 2, kernel=1)
    st = tracer.stats()
    assert st["lastKernel"] == 1 and st["lastSampleLanes"] == (16 if rays >= 16 else 4 if rays >= 4 else 1)
    assert_bitwise(last, want_last, f"philox {scene} {rays} rays: last frame")
    assert_bitwise(acc, want, f"philox {scene} {rays} rays: accum")
    assert st["rays"] == cnt["rays"]


def test_philox_mode_is_served_by_k_stream_whatever_kernel_is_asked_for(rtx, oracle, tracer):
    """kernel = 0 / automatic in Philox mode both run k_stream's Philox instantiation (the only one with the estimator's
    tree); zero rays per pixel draws nothing and goes to k_trace (0 / 0 = NaN in both modes, as in the oracle)."""
    b = _philox(rtx.scenes.mesh_test_scene(48, 40).build_buffers(), 6)
    want, want_last, _ = oracle.render(*b, 0, 3)
    for kernel in (0, -1):
        acc, last = run_gpu(tracer, b, 0, 3, kernel=kernel)
        assert tracer.stats()["lastKernel"] == 1
        assert_bitwise(acc, want, f"philox, kernel option {kernel}")
    pcg, _ = run_gpu(tracer, (rtx.scenes.mesh_test_scene(48, 40).build_buffers()[0],) + b[1:], 0, 3)
    assert not np.array_equal(pcg, acc)
    z = _philox(b, 0)
    acc, last = run_gpu(tracer, z, 0, 1, kernel=-1)
    want, _, _ = oracle.render(*z, 0, 1)
    assert_bitwise(acc, want, "philox, zero rays per pixel")


@pytest.mark.parametrize("rays", [4, 64])
def test_philox_mode_many_frames_in_one_launch_and_every_tuning_knob(rtx, oracle, tracer, rays):
    """Items of several frames in one queue (21 frames, one launch), f16 and f32 nodes, groups of 1 / 5 / 40 items per fetch, a
    two-entry LDS stack (spill path), SHADE thresholds 1 and 64: the accumulated image and the ray count are the oracle twin's."""
    b = _philox(rtx.scenes.mesh_test_scene(80, 56).build_buffers(), rays, maxBounceCount=3)
    nf = 21 if rays == 4 else 3
    want, want_last, cnt = oracle.render(*b, 0, nf)
    for compact, per_fetch, stack, thr in ((1, 16, 31, 48), (0, 1, 4, 1), (1, 5, 31, 64), (1, 40, 9, 24)):
        tracer.set_option("compact_nodes", compact); tracer.set_option("tiles_per_fetch", per_fetch); tracer.set_option("stream_stack", stack)
        try:
            acc, last = run_gpu(tracer, b, 0, nf, kernel=1, shade_threshold=thr)
            st = tracer.stats()
        finally:
            tracer.set_option("compact_nodes", 1); tracer.set_option("tiles_per_fetch", 16); tracer.set_option("stream_stack", 0)
        what = f"philox, {nf} frames in one launch, {rays} rays, compact_nodes={compact}, {per_fetch} items per fetch, stack {stack}, threshold {thr}"
        assert st["lastFramesPerLaunch"] == nf and st["lastSampleLanes"] == (16 if rays >= 16 else 4)
        assert_bitwise(last, want_last, what + ": last frame")
        assert_bitwise(acc, want, what + ": accum")
        assert st["rays"] == cnt["rays"]


def test_philox_mode_rows_bands_and_depth_of_field(rtx, oracle, tracer):
    """Global pixel indices key the stream: a row strip and an interleaved band set equal the same rows of the whole image; depth of
    field on (the defocus draws of block 0 are used); counting build == fast build."""
    m = rtx.scenes.mesh_test_scene(72, 64)
    m.defocusStrength, m.divergeStrength = 60.0, 1.5
    b = _philox(m.build_buffers(), 17)
    full, _ = run_gpu(tracer, b, 2, 2, kernel=1)
    want, _, cnt = oracle.render(*b, 2, 2)
    assert_bitwise(full, want, "philox, depth of field")
    strip, _ = run_gpu(tracer, b, 2, 2, kernel=1, rows=(19, 30))
    assert_bitwise(strip, full[19:49], "philox, row strip")
    tracer.set_bands(1, 3)
    try:
        tracer.set_params(b[0]); tracer.reset_accum(); tracer.render(2, 2)
        got = tracer.read_accum()
    finally:
        tracer.set_rows(0, 64)
    rows = [y for y in range(64) if (y // 8) % 3 == 1]
    assert_bitwise(got, full[rows], "philox, bands 1 of 3")
    tracer.set_params(b[0]); tracer.reset_accum(); tracer.render_counting(2, 2)
    st = tracer.stats()
    assert_bitwise(tracer.read_accum(), full, "philox, counting build")
    assert st["rays"] == cnt["rays"] and st["hits"] == cnt["hits"]


@pytest.mark.parametrize("philox", [0, 1])
def test_stream_kernel_without_triangles(rtx, oracle, tracer, philox):
    """A scene of spheres only takes k_stream's instantiation without traversal code (six waves per SIMD): 19 frames in one launch
    (frame-interleaved items / sample lanes), depth of field on, a row strip, and the counting build — image, ray and hit counts are the
    oracle's in both RNG modes."""
    m = rtx.scenes.config1(88, 72)
    m.numRaysPerPixel = 17
    m.defocusStrength, m.divergeStrength = 40.0, 2.0
    params, spheres, tris, infos = m.build_buffers()
    assert len(tris) == 0 and len(spheres) > 0
    params = params.copy(); params["rngMode"] = philox
    b = (params, spheres, tris, infos)
    acc, last = run_gpu(tracer, b, 3, 19, kernel=1)
    st = tracer.stats()
    want, want_last, cnt = oracle.render(*b, 3, 19)
    assert st["lastKernel"] == 1 and st["numBvhNodes"] == 0
    assert_bitwise(last, want_last, f"spheres only through k_stream, rngMode {philox}: last frame")
    assert_bitwise(acc, want, f"spheres only through k_stream, rngMode {philox}: accum")
    assert st["rays"] == cnt["rays"]
    strip, _ = run_gpu(tracer, b, 3, 19, kernel=1, rows=(13, 40))
    assert_bitwise(strip, acc[13:53], "row strip")
    tracer.set_rows(0, 72); tracer.set_params(params); tracer.reset_accum(); tracer.render_counting(3, 19)
    sc = tracer.stats()
    assert_bitwise(tracer.read_accum(), acc, "counting build")
    assert sc["rays"] == cnt["rays"] and sc["hits"] == cnt["hits"] and sc["sphereTests"] == cnt["rays"] * len(spheres)


def test_multi_frame_launch_equals_frame_by_frame(rtx, oracle, tracer):
    """rt_render(first, n) traces n frames per k_trace launch (work items = (frame, tile)) and accumulates them in order
    afterwards; the result equals n single-frame launches and the oracle (sun at 200x: the clamped running average is
    order-sensitive)."""
    b = rtx.scenes.config1(72, 40).build_buffers()
    tracer.set_option("frame_batch", 1)
    try:
        ref, ref_last = run_gpu(tracer, b, 2, 5)
        assert tracer.stats()["lastFramesPerLaunch"] == 1
    finally:
        tracer.set_option("frame_batch", 0)
    got, got_last = run_gpu(tracer, b, 2, 5)
    assert tracer.stats()["lastFramesPerLaunch"] == 5
    tracer.set_option("frame_batch", 2)                      # 2 + 2 + 1
    try:
        got2, got2_last = run_gpu(tracer, b, 2, 5)
    finally:
        tracer.set_option("frame_batch", 0)
    want, want_last, _ = oracle.render(*b, 2, 5)
    for a, l, what in ((ref, ref_last, "frame by frame"), (got, got_last, "one launch"), (got2, got2_last, "batches of 2")):
        assert_bitwise(l, want_last, f"{what}: last frame")
        assert_bitwise(a, want, f"{what}: accum")


def test_automatic_kernel_choice_is_transparent(rtx, oracle, tracer):
    """kernel = -1 (default): frames 0..5 are traced by a mix of k_trace / k_stream launches while the library measures
    which is faster; the image is the oracle's, and the choice is made after three or four frames (a variant's very first launch
    in a context is not used as its timing)."""
    b = rtx.scenes.mesh_test_scene(96, 64).build_buffers()
    acc, last = run_gpu(tracer, b, 0, 6, kernel=-1)
    st = tracer.stats()
    assert st["autoKernel"] in (0, 1)
    want, want_last, cnt = oracle.render(*b, 0, 6)
    assert_bitwise(last, want_last, "auto kernel: last frame")
    assert_bitwise(acc, want, "auto kernel: accum")
    assert st["rays"] == cnt["rays"]
    tracer.render_frame(6)                     # decided: single frames keep using the chosen kernel
    assert tracer.stats()["autoKernel"] == st["autoKernel"]


@pytest.mark.parametrize("size,k", [((93, 61), 2), ((8, 8), 3), ((200, 9), 4), ((5, 130), 2), ((96, 64), 16)])
def test_tile_groups_multi_frame(rtx, oracle, tracer, size, k):
    """k_stream reserving k work items per fetch (lanes flow from one tile of the group to the next without waiting),
    several frames per launch, partial tiles, fewer items than a group: image and ray count are the oracle's."""
    b = rtx.scenes.mesh_test_scene(*size).build_buffers()
    tracer.set_option("tiles_per_fetch", k)
    try:
        acc, last = run_gpu(tracer, b, 1, 3, kernel=1)
        st = tracer.stats()
    finally:
        tracer.set_option("tiles_per_fetch", 16)
    want, want_last, cnt = oracle.render(*b, 1, 3)
    assert_bitwise(last, want_last, f"{size} k={k}: last frame")
    assert_bitwise(acc, want, f"{size} k={k}: accum")
    assert st["rays"] == cnt["rays"]


@pytest.mark.parametrize("stream_tile,size,n,bands", [(2, (93, 61), 9, None), (4, (93, 61), 19, None), (2, (8, 8), 4, None),
                                                      (4, (5, 130), 16, None), (2, (200, 9), 7, None), (4, (96, 64), 33, (1, 3)),
                                                      (2, (96, 70), 6, (0, 2))])
def test_frame_interleaved_sub_tiles(rtx, oracle, tracer, stream_tile, size, n, bands):
    """k_stream with a wave = (4x4 or 2x2 pixels) x (4 or 16 frames of the launch): whole frame groups in one launch, the
    remainder as 8x8 x 1 items; partial tiles, rows in interleaved bands, frame counts that are not multiples of the group.
    Image, last frame and ray count are the oracle's (frames are independent: frag :362; the accumulation stays ordered)."""
    b = rtx.scenes.mesh_test_scene(*size).build_buffers()
    W, H = size
    tracer.set_option("stream_tile", stream_tile)
    try:
        if bands:
            params, spheres, tris, infos = b
            tracer.set_option("kernel", 1)
            tracer.set_params(params)
            tracer.upload(spheres=spheres, triangles=tris, meshinfo=infos)
            tracer.set_bands(*bands)
            tracer.reset_accum()
            tracer.render(2, n)
            acc, last = tracer.read_accum(), tracer.read_last_frame()
            rows = rtx.distributed.band_rows(H, bands[1], bands[0])
        else:
            acc, last = run_gpu(tracer, b, 2, n, kernel=1)
            rows = list(range(H))
        st = tracer.stats()
    finally:
        tracer.set_option("stream_tile", 4)
        tracer.set_rows(0, H)
    want, want_last, cnt = oracle.render(*b, 2, n)
    assert_bitwise(last, want_last[rows], f"stream_tile {stream_tile} {size} x{n}: last frame")
    assert_bitwise(acc, want[rows], f"stream_tile {stream_tile} {size} x{n}: accum")
    if not bands:
        assert st["rays"] == cnt["rays"]


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("mode", [0, 1])
def test_f32_nodes_and_f16_nodes_give_the_same_image(rtx, oracle, tracer, mode, kernel):
    """compact_nodes = 0 (seven f32 plane loads per node visit) against the default f16 form: both equal the oracle."""
    b = rtx.scenes.mesh_test_scene(96, 64).build_buffers()
    want, want_last, cnt = oracle.render(*b, 1, 4, mode=mode)
    for compact in (0, 1):
        tracer.set_option("compact_nodes", compact)
        try:
            acc, last = run_gpu(tracer, b, 1, 4, mode=mode, kernel=kernel)
            st = tracer.stats()
        finally:
            tracer.set_option("compact_nodes", 1)
        assert_bitwise(last, want_last, f"compact_nodes={compact} kernel {kernel} mode {mode}: last frame")
        assert_bitwise(acc, want, f"compact_nodes={compact} kernel {kernel} mode {mode}: accum")
        assert st["rays"] == cnt["rays"]


def _decode_f16_nodes(h):
    """Node4h words [n, 32] -> (mins [n, 3, 4], maxs [n, 3, 4]) as float64 = origin + offset, checking that the plane sets agree."""
    n = h.shape[0]
    halves = h[:, :24].copy().view(np.float16).astype(np.float64).reshape(n, 6, 8)          # six 16-byte sets of 8 halves
    org = h[:, 28:31].copy().view(np.float32).astype(np.float64)                             # [n, 3]
    sets = halves[:, :4].reshape(n, 4, 2, 4)                                                 # set c: (x planes, y planes)
    minx, maxx, miny, maxy = sets[:, 0, 0], sets[:, 1, 0], sets[:, 0, 1], sets[:, 2, 1]
    assert np.array_equal(sets[:, 2, 0], minx) and np.array_equal(sets[:, 3, 0], maxx)       # bit 0 of c picks the x planes
    assert np.array_equal(sets[:, 1, 1], miny) and np.array_equal(sets[:, 3, 1], maxy)       # bit 1 the y planes
    minz, maxz = halves[:, 4, :4], halves[:, 4, 4:]
    assert np.array_equal(halves[:, 5, :4], maxz) and np.array_equal(halves[:, 5, 4:], minz)
    mins = np.stack([minx, miny, minz], 1) + org[:, :, None]
    maxs = np.stack([maxx, maxy, maxz], 1) + org[:, :, None]
    return mins, maxs


@pytest.mark.parametrize("scene", ["mesh_test", "config3", "far"])
def test_f16_boxes_contain_the_f32_boxes(rtx, tracer, scene):
    """Every child box of the f16 node form contains the padded f32 box it was derived from (so the hierarchy still only
    prunes), is at most ~0.1 % of the node's extent wider, holds no f16 denormal, and keeps the child references."""
    if scene == "config3":
        b = rtx.scenes.config3(64, 36).build_buffers()
    else:
        b = rtx.scenes.mesh_test_scene(32, 24).build_buffers()
        if scene == "far":                       # the whole scene 1e5 units from the origin: offsets stay small, the origin carries it
            params, spheres, tris, infos = (a.copy() for a in b)
            off = np.float32([1.0e5, -3.0e4, 7.0e4])
            for k in ("posA", "posB", "posC"):
                tris[k] += off
            infos["boundsMin"] += off; infos["boundsMax"] += off
            params["worldSpaceCameraPos"] += off
            m = params["camLocalToWorld"].reshape(4, 4); m[:3, 3] += off
            b = (params, spheres, tris, infos)
    run_gpu(tracer, b, 0, 1, kernel=1)
    f32, f16 = tracer.read_bvh()
    assert len(f32) > 0
    planes = f32[:, :24].copy().view(np.float32).astype(np.float64).reshape(-1, 6, 4)
    mins32, maxs32 = planes[:, :3], planes[:, 3:]
    assert np.array_equal(f16[:, 24:28], f32[:, 24:28])
    mins16, maxs16 = _decode_f16_nodes(f16)
    used = f32[:, 24:28] != 0xFFFFFFFF
    u3 = np.broadcast_to(used[:, None, :], mins32.shape)
    assert (mins16[u3] <= mins32[u3]).all() and (maxs16[u3] >= maxs32[u3]).all()
    # empty slots stay un-enterable: min = +inf, max = -inf
    assert np.isposinf(mins16[~u3]).all() and np.isneginf(maxs16[~u3]).all()
    # tightness: the widening is bounded by the f16 step at the offset's magnitude (<= 2^-10 of the node's half extent, plus the
    # smallest normal 2^-14)
    lo = np.where(u3, mins32, np.inf).min(axis=2, keepdims=True); hi = np.where(u3, maxs32, -np.inf).max(axis=2, keepdims=True)
    slack = (hi - lo) * 0.5 * 2.0 ** -9 + 2.0 ** -13
    assert ((mins32 - mins16)[u3] <= np.broadcast_to(slack, mins32.shape)[u3]).all()
    assert ((maxs16 - maxs32)[u3] <= np.broadcast_to(slack, mins32.shape)[u3]).all()
    halves = f16[:, :24].copy().view(np.uint16)
    assert not (((halves & 0x7C00) == 0) & ((halves & 0x03FF) != 0)).any(), "an f16 denormal was stored"


@pytest.mark.parametrize("philox", [0, 1])
def test_queued_frame_submission_equals_rt_render(rtx, oracle, tracer, philox):
    """rt_submit_frame / rt_wait: a host that hands frames in one at a time — the reference's OnRenderImage pattern, RayTracingManager.cs:74-91
    — gets them traced by the library's worker thread in launches of whatever has queued up, accumulated in submission order.  21 frames
    (sun at 200x: the clamped running average is order-sensitive) == rt_render(0, 21) == the oracle; fewer launches than frames; reading in
    the middle is safe; a gap in the frame indices only splits the launch; queue_depth = 1 degenerates to frame by frame."""
    params, spheres, tris, infos = rtx.scenes.config1(72, 40).build_buffers()
    params = params.copy(); params["rngMode"] = philox
    b = (params, spheres, tris, infos)
    ref, ref_last = run_gpu(tracer, b, 0, 21, kernel=-1)
    want, want_last, _ = oracle.render(*b, 0, 21)
    assert_bitwise(ref, want, "rt_render")
    tracer.reset_accum()
    for f in range(9):
        tracer.submit_frame(f)
    part = tracer.read_accum()                       # every other call waits for the queue first
    assert tracer.stats()["numRenderedFrames"] == 9
    for f in range(9, 21):
        tracer.submit_frame(f)
    tracer.wait()
    st = tracer.stats()
    assert st["numRenderedFrames"] == 21 and 2 <= st["queuedLaunches"] <= 21
    assert_bitwise(tracer.read_accum(), want, "queued submission, accum")
    assert_bitwise(tracer.read_last_frame(), want_last, "queued submission, last frame")
    want9, _, _ = oracle.render(*b, 0, 9)
    assert_bitwise(part, want9, "queued submission, read after nine frames")
    # frame by frame through the queue, and a gap in the indices (frames 0..4 then 30..33: two runs, accumulated in that order)
    tracer.set_option("queue_depth", 1)
    try:
        tracer.reset_accum()
        for f in list(range(5)) + list(range(30, 34)):
            tracer.submit_frame(f)
        tracer.wait()
        assert tracer.stats()["queuedLaunches"] == 9
        gap = tracer.read_accum()
    finally:
        tracer.set_option("queue_depth", 64)
    tracer.reset_accum(); tracer.render(0, 5); tracer.render(30, 4)
    assert_bitwise(gap, tracer.read_accum(), "queued submission with a gap == two rt_render calls")


def test_queued_submission_reports_the_error_of_a_queued_launch(rtx, tracer):
    """A launch that fails inside the worker (here: a row strip outside the image) is reported by the next call that waits, once; the
    context works again afterwards."""
    b = rtx.scenes.config1(32, 24).build_buffers()
    run_gpu(tracer, b, 0, 1)
    tracer.set_rows(20, 10)                          # rows [20, 30) of a 24-row image: checked at launch time
    tracer.submit_frame(0); tracer.submit_frame(1)
    with pytest.raises(rtx.RtError, match="queued frame"):
        tracer.wait()
    tracer.set_rows(0, 24)
    tracer.reset_accum()
    tracer.submit_frame(0)
    tracer.wait()
    assert tracer.stats()["numRenderedFrames"] == 1


@pytest.mark.parametrize("size,diverge", [((96, 64), None), ((31, 23), None), ((200, 120), 0.0), ((64, 48), 40.0)])
@pytest.mark.parametrize("philox", [0, 1])
def test_camera_ray_candidate_lists_do_not_change_the_image(rtx, oracle, tracer, size, diverge, philox):
    """Option primary_lists (csrc/rt_primary.hpp): every pixel's camera rays start from the <= 4 BVH leaves
    that can hold their closest hit instead of from the root.  Image, last frame and ray count with the lists == without == the oracle:
    pixels much larger than the triangles (31x23), a footprint of zero size (DivergeStrength 0), one of many pixels (40), both RNG modes;
    the lists are really in use (statistics), and a camera move or a scene change builds them again."""
    m = rtx.scenes.mesh_test_scene(*size)
    if diverge is not None:
        m.divergeStrength = diverge
    params, spheres, tris, infos = m.build_buffers()
    params = params.copy(); params["rngMode"] = philox
    b = (params, spheres, tris, infos)
    want, want_last, cnt = oracle.render(*b, 1, 3)
    tracer.set_option("primary_lists", 0)
    try:
        off, off_last = run_gpu(tracer, b, 1, 3, kernel=1)
        builds0 = tracer.stats()["primaryListBuilds"]
    finally:
        tracer.set_option("primary_lists", 1)
    on, on_last = run_gpu(tracer, b, 1, 3, kernel=1)
    st = tracer.stats()
    assert st["primaryListBuilds"] == builds0 + 1 and sum(st["primaryLists"]) == size[0] * size[1]
    assert_bitwise(off, want, "without candidate lists")
    assert_bitwise(on, want, "with candidate lists: accum")
    assert_bitwise(on_last, want_last, "with candidate lists: last frame")
    assert st["rays"] == cnt["rays"]
    if diverge is None and size == (96, 64):
        assert st["primaryLists"][0] + st["primaryLists"][1] + st["primaryLists"][2] > 0.5 * size[0] * size[1]     # most pixels have a list
        tracer.render(4, 2)                                  # same camera, same scene: the lists stay
        assert tracer.stats()["primaryListBuilds"] == builds0 + 1
        p2 = params.copy(); p2["worldSpaceCameraPos"] = params["worldSpaceCameraPos"] + np.float32([0.25, 0.0, 0.0])
        mm = p2["camLocalToWorld"].copy(); mm[3] += np.float32(0.25); p2["camLocalToWorld"] = mm
        tracer.set_params(p2); tracer.reset_accum(); tracer.render(1, 3)
        assert tracer.stats()["primaryListBuilds"] == builds0 + 2
        moved, _, _ = oracle.render(p2, spheres, tris, infos, 1, 3)
        assert_bitwise(tracer.read_accum(), moved, "candidate lists after a camera move")
        tracer.render_frame(4); tracer.set_params(params); tracer.reset_accum(); tracer.render_frame(0)     # back to the first camera: built again
        assert tracer.stats()["primaryListBuilds"] == builds0 + 3
