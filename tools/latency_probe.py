#!/usr/bin/env python3
"""Single-frame latency of rt_render_frame under tuning options: wall and kernel time (diagnostic; run on the GPU box).
usage: tools/latency_probe.py [--rng philox] [--config 3] name=value[,name=value...] ..."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg
rtx = rtx_pkg.load()

ap = argparse.ArgumentParser()
ap.add_argument("--rng", default="pcg")
ap.add_argument("--config", type=int, default=3)
ap.add_argument("sets", nargs="*", default=[""])
a = ap.parse_args()
mgr = {2: rtx.scenes.config2, 3: rtx.scenes.config3, 4: rtx.scenes.config4, 5: rtx.scenes.config5}[a.config]()
params, spheres, tris, infos = mgr.build_buffers()
params = params.copy(); params["rngMode"] = 1 if a.rng == "philox" else 0
with rtx.Tracer(0) as t:
    t.set_params(params); t.upload(spheres=spheres, triangles=tris, meshinfo=infos)
    t.render(0, 4)
    for s in a.sets:
        opts = dict(kv.split("=") for kv in s.split(",") if kv)
        for k, v in opts.items():
            t.set_option(k, int(v))
        t.reset_accum(); t.render_frame(0); t.render_frame(1)
        wall, kern = [], []
        for f in range(2, 10):
            t0 = time.perf_counter(); t.render_frame(f); wall.append(time.perf_counter() - t0)
            kern.append(t.stats()["lastKernelMs"])
        st = t.stats()
        print(json.dumps({"rng": a.rng, "opts": s, "wall_ms": round(1e3 * sorted(wall)[len(wall) // 2], 3), "kernel_ms": round(sorted(kern)[len(kern) // 2], 3),
                          "kernel": st["lastKernel"], "lanes": st["lastSampleLanes"]}), flush=True)
