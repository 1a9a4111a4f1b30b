"""Wave residency by region of k_stream's persistent loop (diagnostic build with cycle stamps at the region boundaries, counting kernel):
share of the waves' time in traversal bursts, SHADE passes and the fetch / idle branch.  RTX_LIB must point at the instrumented build."""
import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import rtx_pkg
rtx = rtx_pkg.load()
for cfg, rng in ((3, 0), (3, 1), (5, 0)):
    mgr = {3: rtx.scenes.config3, 5: rtx.scenes.config5}[cfg]()
    params, spheres, tris, infos = mgr.build_buffers()
    params = params.copy(); params["rngMode"] = rng
    with rtx.Tracer(0) as t:
        t.set_params(params); t.upload(spheres=spheres, triangles=tris, meshinfo=infos)
        t.set_option("kernel", 1)
        t.render(0, 4); t.reset_accum()
        t.render_counting(0, 16)
        st = t.stats()
    L, E = st["phaseLanes"], st["phaseExecs"]
    burst, shade, idle = L[3], L[4], E[3]
    tot = burst + shade + idle
    print(json.dumps({"config": cfg, "rng": rng, "burst": round(burst / tot, 4), "shade": round(shade / tot, 4), "fetch_idle": round(idle / tot, 4),
                      "node_execs_per_ray64": round(E[0] * 64 / st["rays"], 2), "tri_execs": round(E[1] * 64 / st["rays"], 2), "shade_execs": round(E[2] * 64 / st["rays"], 2)}))
