#!/usr/bin/env python3
"""One-off validation of the whole headline job: 1920x1080, 100,440 triangles, 64 rays/pixel/frame x 16 frames (1024 spp),
8 bounces — the accumulated resultTexture of the GPU against the CPU oracle (through the oracle's search tree; about four
minutes of 128 host threads).  Prints one line per frame and the verdict; tests/ hold the single-frame version of this."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rtx_pkg

rtx = rtx_pkg.load()
import oracle_binding

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
philox = len(sys.argv) > 2 and sys.argv[2] == "philox"          # the counter-based mode against its oracle twin
orc = oracle_binding.Oracle()
m = rtx.scenes.config3()
b = list(m.build_buffers())
if philox:
    p = b[0].copy(); p["rngMode"] = 1; b[0] = p
t = rtx.Tracer(0)
t.set_params(b[0]); t.upload(spheres=b[1], triangles=b[2], meshinfo=b[3])
t0 = time.time(); t.render(0, frames); gpu_s = time.time() - t0
got = t.read_accum(); st = t.stats()
acc, rays = None, 0
for f in range(frames):
    t1 = time.time()
    cur, cnt = orc.render_frame(*b, f, accel=True)
    if acc is None:
        acc = np.zeros_like(cur)
    orc.accumulate(acc, cur, f)
    rays += cnt["rays"]
    print(f"frame {f}: oracle {cnt['rays']:,} rays in {time.time() - t1:.1f} s", flush=True)
same = (got.view(np.uint32) == acc.view(np.uint32)) | (np.isnan(got) & np.isnan(acc))
print(f"mode {'PHILOX' if philox else 'PCG'}, kernel {st['lastKernel']}, sample lanes {st['lastSampleLanes']}; GPU: {st['rays']:,} rays, {frames} frames in {gpu_s:.2f} s (with BVH build and calibration); oracle: {rays:,} rays")
print("VERDICT:", "bit-identical" if same.all() and rays == st["rays"] else f"{int((~same).any(-1).sum())} pixels differ")
sys.exit(0 if same.all() else 1)
