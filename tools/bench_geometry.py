#!/usr/bin/env python3
"""Times the on-device geometry pipeline (transform + chunk bounds + re-layout + BVH refit) against the reference's way
(host transform of every triangle + full re-upload + BVH rebuild), for the 100k- and 1M-triangle workloads."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rtx_pkg

rtx = rtx_pkg.load()
BYTES_PER_TRI = 72 + 72 + 4 + 72 + 72 + 96 + 8 + 72     # transform R/W, bounds R, re-layout R/W, refit leaf R


def main():
    out = []
    tr = rtx.Tracer(0)
    opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    for k, v in opts.items():
        tr.set_option(k, int(v))
    for name, gen in (("config3", rtx.scenes.config3), ("config5", rtx.scenes.config5)):
        mgr = gen(256, 144)
        t0 = time.perf_counter()
        params, spheres, tris, infos = mgr.build_buffers()
        t_marshal = time.perf_counter() - t0
        tr.set_params(params)
        tr.set_rows(0, 144)
        t0 = time.perf_counter()
        tr.upload(spheres=spheres, triangles=tris, meshinfo=infos)
        tr.render(0, 0 + 1)
        t_upload_build_render = time.perf_counter() - t0
        # device pipeline
        tr.upload(spheres=spheres)
        tr.upload_local_meshes(*mgr.build_local_buffers(), len(mgr.meshes))
        xf = mgr.build_transforms()
        tr.set_mesh_transforms(xf)
        tr.render(0, 1)                                   # first use: topology build
        ms = []
        for i in range(5):
            xf["position"][:, 1] += np.float32(0.01)      # move everything a little
            tr.set_mesh_transforms(xf)
            t0 = time.perf_counter()
            tr.read_world_geometry() if False else None
            tr.render(1, 0)                               # n_frames = 0: geometry update only
            wall = time.perf_counter() - t0
            ms.append((tr.stats()["lastGeometryMs"], wall * 1e3))
        gms = float(np.median([m[0] for m in ms]))
        nt = len(tris)
        nodes = tr.stats()["numBvhNodes"]
        alg = nt * BYTES_PER_TRI + nodes * 128 * 3
        out.append({"workload": name, "options": opts, "triangles": nt, "meshes": len(mgr.meshes), "bvh_nodes": nodes,
                    "device_geometry_ms": round(gms, 4), "device_wall_ms_incl_40B_per_mesh_upload": round(float(np.median([m[1] for m in ms])), 3),
                    "algorithmic_bytes": alg, "achieved_GBps": round(alg / (gms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(alg / (gms * 1e-3) / 8e12, 4),
                    "reference_way_host_transform_s": round(t_marshal, 3), "reference_way_upload_plus_bvh_build_s": round(t_upload_build_render, 3)})
    tr.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
