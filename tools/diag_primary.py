import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import rtx_pkg
rtx = rtx_pkg.load()
import numpy as np
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
    if len(sys.argv) > 1 and sys.argv[1] == "top":
        print(json.dumps({"config": cfg, "rng": rng, "nodes_per_ray": st["nodeVisits"] / st["rays"], "share_of_node_lane_visits_top_N": L[3] / max(L[0], 1),
              "share_top_4N+1": L[4] / max(L[0], 1), "node_execs": E[0], "execs_all_lanes_top_N": E[3] / max(E[0], 1), "execs_all_lanes_top_4N+1": E[4] / max(E[0], 1)}))
        continue
    print(json.dumps({"config": cfg, "rng": rng, "rays": st["rays"], "nodes": st["nodeVisits"], "tris": st["triTests"],
          "primary_node_lane_share": L[3] / max(L[0], 1), "primary_tri_lane_share": L[4] / max(L[1], 1),
          "node_execs": E[0], "node_execs_with_primary": E[3], "tri_execs": E[1], "tri_execs_with_primary": E[4],
          "primary_nodes_per_primary_ray": L[3] / (1920 * 1080 * 64 * 16), "hits": st["hits"]}))
