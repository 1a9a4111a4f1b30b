#!/usr/bin/env python3
"""bench.py — headline benchmark of the per-pixel ray-trace path on MI355X.

Metric (BASELINE.json): Mrays/s and ms/frame at 1920x1080, 64 rays/pixel/frame (1024 spp = 16 frames), 8 bounces,
on the ~100k-triangle scene (configs[2]).  A *step* is one frame = one trace+accumulate pass over the image
(RayTracingManager.OnRenderImage, RayTracingManager.cs:74-81); a *ray* is one CalculateRayCollision
(RayTracing.shader:256), counted by the kernel itself.

    python bench.py --gpus 1 --steps 16 --warmup 1
    python bench.py --gpus N --steps K --warmup W        (starts the N ranks itself, as a child process, and relays their one line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

N > 1: the image's 8-row bands are dealt round-robin to the N ranks (row-strip decomposition, interleaved so that sky
rows and object rows spread evenly; seeds use global pixel coordinates, so the image does not depend on the
decomposition), every rank traces its rows for all K frames, and one RCCL gather to rank 0 at the end collects the
accumulated rows (inside the timed region).  Total work is fixed -> "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
PEAK_VALU_TFLOPS = 157.3     # MI355X_MICROARCH.md: FP32 vector peak = 256 CUs x 4 SIMDs x 1 wave64 FMA per 2 cycles x 2.4 GHz
PEAK_VALU_WAVE_INSTR = 256 * 4 * 2.4e9 / 2       # wave-instructions per second at that rate
# algorithmic bytes per ray = nodes x 32 + triangles x 48 + spheres x 16 + (hit ? 64 + 48 : 0); per pixel and frame 48.
# The survey's 32 B assumed a binary node; the BVH4 node this build visits is loaded as 5 x 16 B (f16 form) or 7 x 16 B (f32 form).
NODE_BYTES_SURVEY, NODE_BYTES_F16, NODE_BYTES_F32 = 32, 80, 112
TRI_BYTES, SPHERE_BYTES, HIT_BYTES, PIXEL_BYTES = 48, 16, 64 + 48, 48
VMEM_CYCLES_PER_WAVE_LOAD = 16.2     # tools/ubench/vmem_rate.hip (profiles/ubench_r02_vmem_rate.txt): L1-resident dwordx2/x4, per CU


def csrc_sha16():
    """fingerprint of the kernel sources (csrc/ + include/rt.h + the build flags of __graft_entry__.py): stamps the committed PMC passes (profiles/pmc_table.json), so that a
    pass taken on other kernels than the ones running is flagged instead of silently multiplied with a live time"""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "ray-tracing-extended_amd", "csrc")
    for f in sorted(os.listdir(base)):
        if f.endswith((".hpp", ".hip", ".cpp", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(base, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "rt.h"), "rb").read())
    try:                                    # the compiler flags are part of what runs (round 4: two -mllvm flags were worth 4.6 %)
        import __graft_entry__ as g
        h.update(" ".join(list(g.HIPCC_FLAGS) + list(g.STREAM_TU_FLAGS)).encode())
    except Exception:
        pass
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, help="workload: 2 (12 spheres), 3 (~100k tris, headline), 4 (same scene at 3840x2160, 12 bounces), 5 (~1M tris, DOF)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--rays", type=int, default=0, help="override rays per pixel per frame (default 64)")
    ap.add_argument("--kernel", type=int, default=-2, help="tuning: -1 automatic (library default), 0 k_trace, 1 k_stream")
    ap.add_argument("--shade-threshold", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="name=value tuning option passed to rt_set_option")
    ap.add_argument("--decomposition", choices=["bands", "strips"], default="bands",
                    help="N>1: interleaved 8-row bands (balanced, default) or N contiguous strips")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--rng", choices=["pcg", "philox"], default="pcg", help="pcg = the reference's stream (parity mode, headline); philox = the counter-based latency mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the (untimed) counting pass")
    ap.add_argument("--no-latency", action="store_true", help="skip the (untimed) single-frame latency measurements")
    ap.add_argument("--as-rank-of", type=int, default=0, metavar="N",
                    help="diagnostic on one GPU: render only what rank 0 of N ranks would (bands 0, N, 2N, ...): the compute side of the N-GPU "
                         "strong-scaling run without the gather; the printed value is this rank's own rate, not a job rate")
    ap.add_argument("--rehearse-comm", action="store_true",
                    help="no device, no rendering: the ranks only rendezvous, gather analytically filled strips and build the job report — "
                         "what a box without a GPU can check of the N > 1 launch path (value = 0, data = 'none')")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): start the N ranks with torch.distributed.run as a CHILD
    process — before this process has imported torch or touched a device — relay rank 0's JSON line and the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)        # (stderr goes straight through)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode == 0 and len(lines) != 1:
        sys.stdout.write(r.stdout)
        raise SystemExit(f"the {args.gpus}-rank run printed {len(lines)} JSON lines, expected one")
    for ln in (lines if r.returncode == 0 else r.stdout.splitlines()):
        print(ln, flush=True)
    raise SystemExit(r.returncode)


def rehearse_comm(args, world, rank):
    """--rehearse-comm: everything of the N > 1 path that needs no device — rendezvous, the banded gather of strips whose content is a
    function of (global row, column), the job report — checked on rank 0 and printed as a line of the usual shape (value 0: nothing was traced)."""
    import torch
    import torch.distributed as dist
    import rtx_pkg
    rtx = rtx_pkg.load()
    dist.init_process_group(args.backend if args.backend != "nccl" else "gloo")
    W, H = args.width or 64, args.height or 43
    rows = rtx.distributed.band_rows(H, world, rank)
    per = rtx.distributed.band_rows_padded(H, world)
    strip = torch.zeros(per, W, 4)
    for i, y in enumerate(rows):
        strip[i] = (y * W + torch.arange(W, dtype=torch.float32))[:, None] * torch.tensor([1.0, 2.0, 3.0, 0.0]) + torch.tensor([0.0, 0.0, 0.0, 1.0])
    dist.barrier()
    t0 = time.perf_counter()
    image = rtx.distributed.gather_image_banded(strip, H, dist)
    gather_ms = (time.perf_counter() - t0) * 1e3
    dist.barrier()
    dt = time.perf_counter() - t0
    job = rtx.distributed.job_report(dist, "cpu", "gloo", 0.0, dt, 0.0, gather_ms, strip.numel() * 4)
    if job["world_seen"] != world:
        raise SystemExit(f"the process group has {job['world_seen']} ranks, WORLD_SIZE says {world}")
    if rank == 0:
        want = (torch.arange(H * W, dtype=torch.float32).reshape(H, W))
        if not (torch.equal(image[..., 0], want) and torch.equal(image[..., 2], want * 3.0) and bool((image[..., 3] == 1).all())):
            raise SystemExit("rehearsal: the gathered bands are not in their places")
        print(json.dumps({"metric": "Mrays/s", "value": 0.0, "unit": "Mrays/s", "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": 0.0,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "none",
                          "config": {"workload": f"communication rehearsal without a device: {W}x{H} image in interleaved 8-row bands over {world} ranks, no rendering"},
                          "per_rank": job["per_rank"], "comm": job["comm"], "roofline": None, "cpu_baseline": None}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def build_workload(rtx, args):
    gen = {2: rtx.scenes.config2, 3: rtx.scenes.config3, 4: rtx.scenes.config4, 5: rtx.scenes.config5}[args.config]
    mgr = gen(args.width, args.height) if args.width and args.height else gen()
    if args.rays:
        mgr.numRaysPerPixel = args.rays
    return mgr


def cpu_baseline(rtx, buffers):
    """The oracle (reference algorithm: flat chunk loop) timed on this box's host cores on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    orc = oracle_binding.Oracle()
    params, spheres, tris, infos = buffers
    W, H = int(params["width"]), int(params["height"])
    p = params.copy()
    rays_pp = int(p["numRaysPerPixel"])
    # calibrate on a 16x16 window, then size the sample (crop side, then rays/pixel) for ~15 s of wall time
    t0 = time.time()
    orc.render_frame(p, spheres, tris, infos, 0, ((W - 16) // 2, (H - 16) // 2, (W + 16) // 2, (H + 16) // 2))
    per_px = max(time.time() - t0, 1e-3) / 256
    n = 64
    while n < 512 and per_px * (2 * n) ** 2 < 15:
        n *= 2
    n = min(n, W, H)
    if per_px * n * n > 30:
        p["numRaysPerPixel"] = max(1, int(rays_pp * 20 / (per_px * n * n)))
    x0, y0 = (W - n) // 2, (H - n) // 2
    frames = max(1, min(64, int(10.0 / max(per_px * n * n, 1e-3))))       # cheap scenes: several frames, ~10 s in all
    rays, t0 = 0, time.time()
    for f in range(frames):
        _, c = orc.render_frame(p, spheres, tris, infos, f, (x0, y0, x0 + n, y0 + n))
        rays += c["rays"]
    dt = time.time() - t0
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": c["threads"], "kind": "port",
           "sample": f"{n}x{n} centre crop of frame(s) 0..{frames - 1}, {int(p['numRaysPerPixel'])} rays/pixel, FLAT_CHUNKS "
                     f"(the reference's chunk loop), {rays} rays in {dt:.1f} s"}
    # beside it: the same oracle finding triangles through its own search tree (not the reference's algorithm — the fair
    # CPU comparison for a BVH tracer), ~8 s on a larger crop at the workload's own rays per pixel
    p = params.copy()
    t0 = time.time()
    _, c = orc.render_frame(p, spheres, tris, infos, 0, ((W - 64) // 2, (H - 64) // 2, (W + 64) // 2, (H + 64) // 2), accel=True)
    per_px = max(time.time() - t0, 1e-3) / 4096             # includes the tree build: an upper bound
    n = 64
    while n < min(W, H) and per_px * (2 * n) ** 2 < 8:
        n *= 2
    n = min(n, W, H)
    x0, y0 = (W - n) // 2, (H - n) // 2
    t0 = time.time()
    _, c = orc.render_frame(p, spheres, tris, infos, 0, (x0, y0, x0 + n, y0 + n), accel=True)
    dt = time.time() - t0
    out["with_search_tree"] = {"value": c["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": c["threads"],
                               "sample": f"{n}x{n} centre crop of frame 0, {rays_pp} rays/pixel, oracle search tree "
                                         f"(build included), {c['rays']} rays in {dt:.1f} s"}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)                        # never returns: one child process runs the N ranks
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match the {world} rank(s) this process was launched as (WORLD_SIZE)")
    if args.rehearse_comm:
        return rehearse_comm(args, world, rank)
    import numpy as np
    import torch
    import rtx_pkg
    rtx = rtx_pkg.load()

    dist = None
    dev_index = int(os.environ.get("RTX_BENCH_DEVICE", local_rank))      # rehearsal on a 1-GPU box: all ranks on device 0
    # RTX_BENCH_FORCE_DIST=1 (tests, one-GPU box): take the N > 1 code path with a process group of ONE rank — RCCL initialised through
    # torch.distributed, barrier, the frame-end gather and the job report all run for real, on one device
    force_dist = os.environ.get("RTX_BENCH_FORCE_DIST") == "1" and world == 1
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    comm_dev = f"cuda:{dev_index}" if args.backend == "nccl" else "cpu"

    mgr = build_workload(rtx, args)
    buffers = mgr.build_buffers()
    params, spheres, tris, infos = buffers
    if args.rng == "philox":
        params = params.copy()
        params["rngMode"] = 1
        buffers = (params, spheres, tris, infos)
    if os.environ.get("RTX_BENCH_BRUTE") == "1":           # diagnostic only (what the literal chunk-box filter costs): NOT the benchmark's semantics
        params = params.copy()
        params["intersectMode"] = 1
        buffers = (params, spheres, tris, infos)
    W, H = int(params["width"]), int(params["height"])
    banded = (world > 1 or force_dist) and args.decomposition == "bands"
    sim = args.as_rank_of if world == 1 and args.as_rank_of > 1 and not force_dist else 0
    if sim:
        my_rows = rtx.distributed.band_rows(H, sim, 0)
        row0, nrows, rows = 0, len(my_rows), rtx.distributed.band_rows_padded(H, sim)
    elif banded:      # 8-row bands dealt round-robin: sky rows and object rows spread evenly over the ranks
        my_rows = rtx.distributed.band_rows(H, world, rank)
        row0, nrows, rows = rank * 8, len(my_rows), rtx.distributed.band_rows_padded(H, world)
    else:
        row0, nrows, rows = rtx.distributed.row_strip(H, world, rank)

    tr = rtx.Tracer(dev_index)
    tr.set_params(params)
    tr.upload(spheres=spheres, triangles=tris, meshinfo=infos)
    if sim:
        tr.set_bands(0, sim)
    elif banded:
        tr.set_bands(rank, world)
    else:
        tr.set_rows(row0, nrows)
    if args.kernel >= -1:
        tr.set_option("kernel", args.kernel)
    if args.shade_threshold:
        tr.set_option("shade_threshold", args.shade_threshold)
    if args.blocks_per_cu:
        tr.set_option("blocks_per_cu", args.blocks_per_cu)
    for kv in args.opt:
        k, v = kv.split("=")
        tr.set_option(k, int(v))
    strip = torch.zeros(rows, W, 4, dtype=torch.float32, device=f"cuda:{dev_index}")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- setup (untimed): upload, BVH build, and the library's calibration frames (tile costs for its costliest-first
    # tile order; k_trace vs k_stream timing for its automatic kernel choice) — then the W warmup steps
    tr.render(0, 4)
    tr.reset_accum()
    tr.render(0, max(args.warmup, 0))
    gather = rtx.distributed.gather_image_banded if banded else rtx.distributed.gather_image
    if dist is not None:                      # warm the RCCL communicator
        gather(strip.to(comm_dev), H, dist)
    # ---- timed region: exactly K steps + the frame-end gather
    tr.reset_accum()
    barrier()
    barrier()                                 # (the collective behind the barrier is warm as well when the timed region starts)
    t0 = time.perf_counter()
    tr.render(0, args.steps)
    render_wall_ms = (time.perf_counter() - t0) * 1e3
    gather_ms = 0.0
    if dist is not None:
        tg = time.perf_counter()
        tr.copy_accum_to_device(strip.data_ptr(), nrows * W * 4)
        image = gather(strip.to(comm_dev), H, dist)   # [H, W, 4] on rank 0
        if comm_dev != "cpu":
            torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
    tb = time.perf_counter()
    barrier()
    dt = time.perf_counter() - t0
    closing_barrier_ms = (time.perf_counter() - tb) * 1e3
    st = tr.stats()
    job = rtx.distributed.job_report(dist, comm_dev, args.backend if dist is not None else None, float(st["rays"]), dt, st["totalKernelMs"],
                                     gather_ms, strip.numel() * 4)
    total_rays, wall = job["total_rays"], job["wall_s"]
    if dist is not None:
        job["comm"]["closing_barrier_ms_rank0"] = round(closing_barrier_ms, 3)     # rank 0's wait in the barrier that ends the timed region
        job["comm"]["render_wall_ms_rank0"] = round(render_wall_ms, 3)             # rank 0: rt_render of the K steps, call to return
    if job["world_seen"] != world:
        raise SystemExit(f"the {args.backend} process group has {job['world_seen']} ranks, WORLD_SIZE says {world}")
    kernel_ms_rank0 = st["totalKernelMs"]
    busy = job["per_rank"]["kernel_ms"]

    # ---- roofline (rank 0's strip): counting build of the same kernel over the same K frames, untimed
    roofline = None
    kernel_names = ({1: "k_stream<false,true,{H}>"} if args.rng == "philox" else
                    {0: "k_trace<false,false,{H}>", 1: "k_stream<false,false,{H}>"})
    chosen = st.get("lastKernel", -1)                    # the kernel that ran the timed launches
    opts = dict(kv.split("=") for kv in args.opt)
    compact = int(opts.get("compact_nodes", 1)) != 0
    kernel_name = kernel_names.get(chosen, "k_trace<false,false,{H}>").replace("{H}", "true" if compact else "false")
    if len(tris) == 0 and kernel_name.startswith("k_trace"):
        kernel_name = "k_trace<false,false,false,6>"                      # spheres only: the six-waves-per-SIMD instantiation
    elif len(tris) == 0 and kernel_name.startswith("k_stream"):
        kernel_name = f"k_stream<false,{'true' if args.rng == 'philox' else 'false'},false,false>"    # ... and k_stream's, without traversal code
    if not args.no_roofline:
        tr.reset_accum()
        tr.render_counting(0, args.steps)
        sc = tr.stats()
        px = nrows * W * args.steps
        fpl = max(1, int(st.get("lastFramesPerLaunch", 1)))           # frames traced per launch in the timed region
        launches = (args.steps + fpl - 1) // fpl if args.steps else 1
        launch_s = kernel_ms_rank0 / 1e3 / max(launches, 1)

        def alg_bytes(node_bytes):
            return (sc["nodeVisits"] * node_bytes + sc["triTests"] * TRI_BYTES + sc["sphereTests"] * SPHERE_BYTES
                    + sc["hits"] * HIT_BYTES + px * PIXEL_BYTES) / max(launches, 1)
        loaded = alg_bytes(NODE_BYTES_F16 if compact else NODE_BYTES_F32)
        # PMC figures of the same workload and kernel (rocprofv3 --pmc passes, profiles/pmc_table.json), stored per frame so that
        # any --steps scales them; only for the N = 1 launch they were measured on
        pmc, traffic = None, None
        tf = os.path.join(ROOT, "profiles", "pmc_table.json")
        if os.path.exists(tf):
            try:
                table = json.load(open(tf))
            except Exception:
                table = {}
            sfx = "_philox" if args.rng == "philox" else ""
            exact = f"config{args.config}_{W}x{H}_{int(params['numRaysPerPixel'])}{sfx}"
            ent = table.get(exact) if (nrows == H and world == 1 and not banded) else None
            if ent is None:
                # the same scene at another resolution / ray count, or a part of the image (a rank of N, --as-rank-of): the per-frame figures
                # of its committed pass, scaled by rays (the instruction and byte counts per ray of a scene do not depend on the image size)
                for key, cand in sorted(table.items(), key=lambda kv: kv[0] != exact):
                    if key.startswith(f"config{args.config}_") and key.endswith("_philox") == bool(sfx) and cand.get("rays_per_frame") and sc["rays"] and args.steps:
                        k = (sc["rays"] / args.steps) / cand["rays_per_frame"]
                        ent = {f: (v * k if f.endswith("_per_frame") and isinstance(v, (int, float)) else v) for f, v in cand.items()}
                        ent["source"] = f"{cand.get('source')}; scaled by rays per frame x{k:.4f} from {key}"
                        break
            if ent and ent.get("kernel", "").split("<")[0] == kernel_name.split("<")[0]:
                pmc = dict(ent)
                pmc["stale"] = ent.get("csrc_sha16") != csrc_sha16()      # the pass ran other kernel sources than the ones timed here
                traffic = int(ent["hbm_bytes_per_frame"] * fpl)
        hbm = {"peak": PEAK_HBM_GBPS, "unit": "GB/s",
               "algorithmic_survey_32B_nodes": round(alg_bytes(NODE_BYTES_SURVEY) / launch_s / 1e9, 1),
               "algorithmic_bytes_loaded": round(loaded / launch_s / 1e9, 1),
               "note": "algorithmic bytes are served by L1/L2/Infinity Cache (hit rates in pmc); `measured` is what reaches HBM",
               "measured": round(traffic / launch_s / 1e9, 1) if traffic else None,
               "measured_frac": round(traffic / launch_s / 1e9 / PEAK_HBM_GBPS, 5) if traffic else None}
        hbm["algorithmic_survey_frac"] = round(hbm["algorithmic_survey_32B_nodes"] / PEAK_HBM_GBPS, 5)
        # What binds k_stream (round 3, profiles/ab_probe_r03.txt + profiles/ubench_r03_node_mix.txt): instruction ISSUE, on two ports at once.
        # One more vector-memory instruction per node step costs 5.2 % (13-15 % for two), thirteen more VALU instructions 2.6 % (4.8 % for
        # 26), on both triangle workloads; the node step's own instruction mix, run in lockstep on L1-resident data with every lane
        # active, issues 0.244 VALU wave-instructions per cycle and SIMD at five waves (344.8 cycles per step of 84 VALU + 5 loads + 4 LDS)
        # against the 0.5 of the spec — most of it is half-rate (f16->f32 FMAs, min/max, compares, selects).  So the line reports
        # the VALU issue rate against the spec AND the share of the launch that the kernel's node steps and triangle tests would take at
        # the micro-benchmark's rates, the vector-memory issue share, and the HBM contract figures of SURVEY 8(d) (cache-served).
        MIX = {"node_step_cycles_per_simd": 338.1, "valu_per_node_step": 88, "triangle_test_cycles_per_simd": 228.8, "valu_per_triangle_test": 54,
               "waves_per_simd": 6, "source": "profiles/node4_ubench_r04.txt / ubench_r04_node_mix.txt (tools/ubench/node_mix.hip: the product's node_step<true> / "
                                               "ray_triangle in lockstep, all lanes active, data in L1, six waves per SIMD; VALU per iteration = SQ_INSTS_VALU of that run)"}
        PROBES = {"plus_1_load_per_node_step": {"config3": -0.052, "config5": -0.052}, "plus_2_loads_per_node_step": {"config3": -0.133, "config5": -0.153},
                  "plus_13_valu_per_node_step": {"config3": -0.026, "config5": -0.028}, "plus_26_valu_per_node_step": {"config3": -0.048},
                  "plus_1_load_per_triangle_test": {"config3": -0.017, "config5": -0.014}, "plus_13_valu_per_triangle_test": {"config3": -0.013, "config5": -0.011},
                  "source": "profiles/ab_probe_r03.txt (A/B builds on one box, bench.py --steps 16; throughput change)"}
        execs = {n: e for n, e in zip(("node", "triangle", "shade", "environment", "camera"), sc["phaseExecs"])}
        simd_cycles = launch_s * launches * 2.4e9 * 1024                     # SIMD cycles of the timed launches at the nominal clock
        roofline = {"bound": "issue", "unit": "Gwave-instr/s", "peak": round(PEAK_VALU_WAVE_INSTR / 1e9, 1), "achieved": None, "frac": None,
                    "traffic": traffic, "kernel": kernel_name, "frames_interleaved_per_wave": st.get("lastFramesInterleaved", 1),
                    "sample_lanes_per_pixel": st.get("lastSampleLanes", 1), "launch_ms": round(launch_s * 1e3, 3), "frames_per_launch": fpl,
                    "launches": launches, "algorithmic_bytes_per_launch": int(loaded),
                    "definition": "bound = instruction issue (VALU and vector-memory ports together, see probes): achieved = VALU wave-instructions/s "
                                  "(SQ_INSTS_VALU per frame from the committed rocprofv3 pass x frames / launch time measured here), peak = the spec's one "
                                  "wave64 instruction per 2 cycles per SIMD; traversal_at_mix_ceiling = share of the launch's SIMD cycles that its node "
                                  "steps and triangle tests take at the rates their instruction mix reaches in the micro-benchmark; the working set is "
                                  "cache-resident, so HBM is not the roof (hbm)",
                    "traversal_at_mix_ceiling": round((execs["node"] * MIX["node_step_cycles_per_simd"] + execs["triangle"] * MIX["triangle_test_cycles_per_simd"])
                                                      / simd_cycles, 4) if kernel_name.startswith("k_stream") else None,
                    "mix_ceiling": MIX, "probes": PROBES,
                    "hbm": hbm,
                    "per_ray": {"nodes": round(sc["nodeVisits"] / max(sc["rays"], 1), 2),
                                "tris": round(sc["triTests"] / max(sc["rays"], 1), 2),
                                "spheres": round(sc["sphereTests"] / max(sc["rays"], 1), 2)},
                    "phase_lane_utilisation": {n: round(l / max(64 * e, 1), 3) for n, l, e in
                                               zip(("node", "triangle", "shade", "environment", "camera"), sc["phaseLanes"], sc["phaseExecs"])},
                    "phase_wave_execs_per_ray": {n: round(e * 64 / max(sc["rays"], 1), 2) for n, e in
                                                 zip(("node", "triangle", "shade", "environment", "camera"), sc["phaseExecs"])},
                    "primitive_tests_per_s": {"box": round(sc["nodeVisits"] * 4 / (kernel_ms_rank0 / 1e3), 0),
                                              "triangle": round(sc["triTests"] / (kernel_ms_rank0 / 1e3), 0),
                                              "sphere": round(sc["sphereTests"] / (kernel_ms_rank0 / 1e3), 0)},
                    "bvh": {"nodes": sc["numBvhNodes"], "max_stack": sc["bvhMaxStack"]},
                    "camera_ray_lists": {"pixels_bounded_list": int(sc["primaryLists"][0]), "pixels_unbounded_list": int(sc["primaryLists"][1]),
                                         "pixels_certain_miss": int(sc["primaryLists"][2]), "pixels_no_list": int(sc["primaryLists"][3]),
                                         "build_ms": round(sc["lastPrimaryListsMs"], 3), "builds": int(sc["primaryListBuilds"]),
                                         "note": "per pixel, the <= 4 BVH leaves that can hold the closest hit of its camera rays (csrc/rt_primary.hpp), built once per "
                                                 "camera / scene before the warm-up — an acceleration structure like the BVH, outside the timed region; every ray is still traced"},
                    "pmc": None}
        # ---- the launch's VALU instruction count from THIS run's own counting pass: wave-level executions of every region of k_stream
        # (rt_stats.phaseExecs / schedExecs) x the region's static VALU count from the code object's assembly (tools/static_valu.py ->
        # static_valu.json, stamped with the source hash).  An independent figure next to the stored rocprofv3 one: they must agree.
        valu_model = None
        sv_path = os.path.join(ROOT, "ray-tracing-extended_amd", "static_valu.json")
        if kernel_name.startswith("k_stream") and os.path.exists(sv_path):
            try:
                sv = json.load(open(sv_path))
            except Exception:
                sv = {}
            key = kernel_name.replace("k_stream<false,false,true>", "k_stream<false,false,true,true>").replace("k_stream<false,true,true>", "k_stream<false,true,true,true>") \
                             .replace("k_stream<false,false,false>", "k_stream<false,false,false,true>").replace("k_stream<false,true,false>", "k_stream<false,true,false,true>")
            ent_sv = (sv.get("kernels") or {}).get(key)
            if ent_sv and sc.get("regionExecs"):
                v = ent_sv["valu"]
                names = sv.get("regions") or []
                ex = {n: float(sc["regionExecs"][i]) for i, n in enumerate(names) if i < len(sc["regionExecs"])}
                ex["loop"] = ex.get("shade", 0.0) + ex.get("burst", 0.0) + ex.get("fetch", 0.0)          # one pass of the persistent loop's head and latch per region entered
                ex["prologue"] = ex["epilogue"] = 0.0                                                    # once per wave: a few thousand waves per launch
                dof_on = not (float(params["defocusStrength"]) == 0.0)                                   # (the host's fixed_origin decision, wave-uniform in the kernel)
                ex["camera_dof"] = ex.get("camera", 0.0) if dof_on else 0.0
                ex["camera_focus"] = 0.0 if int(sc["primaryLists"][0] + sc["primaryLists"][1] + sc["primaryLists"][2] + sc["primaryLists"][3]) > 0 and not dof_on else ex.get("camera", 0.0)
                sun_off = float(params["sunIntensity"]) == 0.0 and not np.signbit(params["sunIntensity"]) and 1.0 <= float(params["sunFocus"]) <= 1.0e6
                ex["env_sun"] = 0.0 if (sun_off or not int(params["environmentEnabled"])) else ex.get("env", 0.0)   # (a wave-uniform branch on the parameters)
                total = sum(v.get(n, 0) * ex.get(n, 0.0) for n in names)
                valu_model = {"valu_wave_instructions_per_frame": round(total / max(args.steps, 1), 0),
                              "regions_static_valu": v, "region_execs_per_frame": {n: round(ex.get(n, 0.0) / max(args.steps, 1), 1) for n in names},
                              "static_csrc_sha16": sv.get("csrc_sha16"), "static_stale": sv.get("csrc_sha16") != csrc_sha16(),
                              "note": "sum over k_stream's regions (csrc/rt_kernels.hpp RT_REGION_LIST) of (wave-level executions counted by this run's counting pass, "
                                      "rt_stats.regionExecs) x (VALU instructions of the region in the code object's assembly, tools/static_valu.py; blocks behind "
                                      "cold-path markers left out)"}
        if pmc:
            instr_s = pmc["valu_wave_instructions_per_frame"] * fpl / launch_s
            roofline["achieved"] = round(instr_s / 1e9, 2)
            roofline["frac"] = round(instr_s / PEAK_VALU_WAVE_INSTR, 5)
            roofline["tflops_equivalent"] = {"achieved": round(instr_s * 64 * 2 / 1e12, 2), "peak": PEAK_VALU_TFLOPS,
                                             "note": "every VALU instruction counted as one FMA per lane: a reference line, not flops"}
            lane = pmc.get("valu_lane_utilisation")
            loads_s = pmc["vmem_read_wave_instructions_per_frame"] * fpl / launch_s
            roofline["frac_active_lanes"] = round(roofline["frac"] * lane, 5) if lane else None
            roofline["vector_memory"] = {"wave_loads_per_s": round(loads_s, 0), "cycles_per_wave_load_per_cu": VMEM_CYCLES_PER_WAVE_LOAD,
                                         "frac": round(loads_s * VMEM_CYCLES_PER_WAVE_LOAD / (256 * 2.4e9), 5)}
            roofline["pmc"] = {k: pmc.get(k) for k in ("l2_hit_rate", "l1_hit_rate", "valu_lane_utilisation", "ta_busy_frac", "td_busy_frac", "wait_any_frac_of_wave_cycles",
                                                        "wait_inst_any_frac_of_wave_cycles", "csrc_sha16", "stale", "source")}
            if valu_model:
                ratio = valu_model["valu_wave_instructions_per_frame"] / max(pmc["valu_wave_instructions_per_frame"], 1.0)
                valu_model["pmc_valu_wave_instructions_per_frame"] = round(pmc["valu_wave_instructions_per_frame"], 0)
                valu_model["model_over_pmc"] = round(ratio, 4)
                valu_model["agree_within_5_percent"] = bool(abs(ratio - 1.0) <= 0.05)
                roofline["pmc"]["stale"] = bool(roofline["pmc"]["stale"] or not valu_model["agree_within_5_percent"])     # a stored count that the run's own counters contradict is stale
        roofline["valu_model"] = valu_model
        if not pmc:
            # no PMC pass for this configuration / kernel: the instruction counts are unknown, so no issue fraction is claimed; the HBM contract
            # figures of SURVEY 8(d) stay under `hbm` (cache-served: they can exceed the HBM peak and are not a roofline fraction)
            roofline["definition"] = ("no committed rocprofv3 --pmc pass for this configuration: achieved / frac are not claimed (tools/profile.sh + "
                                      "tools/make_pmc_table.py add one); hbm = SURVEY 8(d) algorithmic bytes / launch time, cache-served")

    # ---- latency of one rt_render_frame, camera fixed and camera moved before every frame (untimed extras, rank 0, N = 1)
    latency = None
    if world == 1 and not args.no_latency and not sim and not force_dist:
        tr.reset_accum()
        tr.render_frame(0)
        t0 = time.perf_counter()
        for f in range(1, 6):
            tr.render_frame(f)
        single = (time.perf_counter() - t0) / 5
        moved = []
        p2 = params.copy()
        t0 = time.perf_counter()
        for f in range(8):
            p2["worldSpaceCameraPos"] = params["worldSpaceCameraPos"] + np.float32([0.002 * (f + 1), 0.0, 0.0])
            m = p2["camLocalToWorld"].copy(); m[3] = params["camLocalToWorld"][3] + np.float32(0.002 * (f + 1)); p2["camLocalToWorld"] = m
            tr.set_params(p2)
            tr.reset_accum()                                   # a moved camera restarts the accumulation (RayTracingManager.Start)
            tr.render_frame(0)
        moving = (time.perf_counter() - t0) / 8
        tr.set_params(params)
        # the same K frames handed in one at a time through the queue (rt_submit_frame ... rt_wait): what a frame-by-frame host — the
        # reference's OnRenderImage pattern — gets instead of the single-frame rate
        tr.reset_accum()
        tr.render(0, 1)                              # (the camera went back: the library re-measures its tile order on one frame)
        tr.reset_accum()
        t0 = time.perf_counter()
        for f in range(args.steps):
            tr.submit_frame(f)
        tr.wait()
        queued = (time.perf_counter() - t0) / max(args.steps, 1)
        queued_launches = tr.stats()["queuedLaunches"]
        latency = {"latency_ms_single_frame": round(single * 1e3, 3), "latency_ms_single_frame_moving_camera": round(moving * 1e3, 3),
                   "queued_ms_per_frame": round(queued * 1e3, 3), "queued_launches": int(queued_launches),
                   "queued_vs_batched": round((wall / max(args.steps, 1)) / queued, 4) if queued > 0 else None,
                   "note": "wall time of one rt_render_frame (one launch, fused accumulate, stream sync); moving camera = rt_set_params with a "
                           "new camera position + rt_reset_accum + rt_render_frame per frame, tile costs re-measured and re-sorted on the "
                           "device every frame; queued = the K frames of the timed region submitted one by one with rt_submit_frame, then rt_wait "
                           "(wall time per frame; queued_vs_batched = its rate against the timed rt_render(0, K))"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(rtx, buffers)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    tr.close()
    if rank != 0:
        return
    names = {2: "configs[1]: 12 spheres Cornell style", 3: "configs[2]: Chess pieces x17 (100,440 triangles, BVH4)",
             4: "configs[3]: Chess pieces x17 (100,440 triangles, BVH4), 4K",
             5: "configs[4]: Chess pieces x170 (1,004,364 triangles, BVH4), DOF on"}
    out = {
        "metric": "Mrays/s", "value": round(total_rays / wall / 1e6, 2), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / max(args.steps, 1) * 1e3, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{names[args.config]}, {W}x{H}, {int(params['numRaysPerPixel'])} rays/pixel/frame "
                               f"(1024 spp = 16 frames), {int(params['maxBounceCount'])} bounces, {args.rng.upper()}, {'BRUTE (diagnostic)' if int(params['intersectMode']) == 1 else 'FLAT_CHUNKS'} semantics",
                   "decomposition": (f"{world} ranks, " + ("interleaved 8-row bands" if banded else "contiguous row strips")
                                     + " + one RCCL gather") if (world > 1 or force_dist) else
                                    (f"diagnostic: rank 0 of {sim} (its interleaved bands only, no gather)" if sim else "single GPU"),
                   "rays_per_frame": int(total_rays / max(args.steps, 1)),
                   "triangles": int(len(tris)), "chunks": int(len(infos)), "spheres": int(len(spheres)),
                   "per_gpu_kernel_ms": busy},
        "per_rank": job["per_rank"], "comm": job["comm"],
        "roofline": roofline, "cpu_baseline": cpu,
    }
    if latency:
        out.update({k: v for k, v in latency.items() if k != "note"})
        out["latency_note"] = latency["note"]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
