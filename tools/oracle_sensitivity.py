#!/usr/bin/env python3
"""How much of a converged image depends on the float semantics the oracle had to freeze?

The reference evaluates its tracer (Assets/Scripts/Shaders/RayTracing.shader) with Unity's shader compiler and a GPU driver: sin, cos,
log, pow and the contraction of a*b+c into FMAs are theirs, unknown here, and the reference holds no image to pin them (DESIGN.md
section 2: "parity unpinned at float level").  This script bounds what that can matter: it renders the reference's six scenes (the
numeric conversions under tests/golden/scenes/) with three builds of the SAME oracle source —

    frozen   the parity oracle: polynomial kernels, no FMA contraction (what the HIP kernels reproduce bit for bit)
    libm     sinf / cosf / logf / exp2f / powf from the host's libm            (oracle/Makefile: librt_oracle_libm.so)
    fma      the frozen kernels with -ffp-contract=fast -mfma                   (librt_oracle_fma.so)

— accumulated like RayTracingManager.OnRenderImage does (Accumulate.shader), and reports per channel the max and mean |difference| of the
accumulated images, next to the Monte-Carlo noise floor (the same build at another first frame index).  CPU only.

    python tools/oracle_sensitivity.py --size 256 --spp 1024 --out profiles/oracle_sensitivity_r03.json
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--rays", type=int, default=64, help="rays per pixel per frame (frames = spp / rays)")
    ap.add_argument("--scenes", default="Balls_Outdoors,Chess,Knight,Reflective_Balls,Suzanne,Thumbnail")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import rtx_pkg
    rtx = rtx_pkg.load()
    from rtx_amd import unity_scene
    import oracle_binding
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "all", "variants"])
    builds = {"frozen": oracle_binding.Oracle(),
              "libm": oracle_binding.Oracle(os.path.join(ROOT, "oracle", "librt_oracle_libm.so")),
              "fma": oracle_binding.Oracle(os.path.join(ROOT, "oracle", "librt_oracle_fma.so"))}
    frames = max(1, args.spp // args.rays)
    report = {"size": args.size, "spp": frames * args.rays, "rays_per_frame": args.rays, "frames": frames, "scenes": {}}
    for name in args.scenes.split(","):
        mgr = unity_scene.load_scene_npz(os.path.join(ROOT, "tests", "golden", "scenes", name + ".npz"), args.size, args.size)
        mgr.numRaysPerPixel = args.rays
        b = mgr.build_buffers()
        images, t0 = {}, time.time()
        for tag, orc in builds.items():
            images[tag], _, cnt = orc.render(*b, 0, frames, accel=True)
        # the noise floor: the same build on other frame indices (other seeds), accumulated with the weights of frames 0 .. n-1
        acc = None
        for f in range(frames):
            cur, _ = builds["frozen"].render_frame(*b, 1000 + f, accel=True)
            if acc is None:
                acc = np.zeros_like(cur)
            builds["frozen"].accumulate(acc, cur, f)
        images["frozen_other_frames"] = acc
        ref = images["frozen"][..., :3]
        row = {"rays": cnt["rays"], "mean_radiance": float(ref.mean()), "seconds": round(time.time() - t0, 1)}
        for tag in ("libm", "fma", "frozen_other_frames"):
            d = np.abs(images[tag][..., :3] - ref)
            row[tag] = {"max_abs": [float(d[..., c].max()) for c in range(3)], "mean_abs": [float(d[..., c].mean()) for c in range(3)],
                        "pixels_differing": int((d.max(-1) > 0).sum()), "pixels_over_1e-5": int((d.max(-1) > 1e-5).sum())}
        report["scenes"][name] = row
        print(name, json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
