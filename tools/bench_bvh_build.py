#!/usr/bin/env python3
"""BVH builders side by side on the 100k- and 1M-triangle workloads: build time, tree size, worst-case stack, traversal work per
ray (counting build of the tracer, 960x540 x 4 rays, frame 0), and that both trees give the same image.  JSON on stdout."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rtx_pkg

rtx = rtx_pkg.load()


def tree_shape(f32):
    """Levels, mean / max leaf depth weighted by triangles, and mean children per node of a read-back BVH4 (rt_read_bvh f32 form)."""
    refs = f32[:, 24:28]
    n = len(refs)
    depth = np.zeros(n, np.int32)
    order = [0]
    level = [0]
    d = 0
    leaf_depth_sum, leaf_tris, max_leaf_depth = 0, 0, 0
    leaf_sizes = [0, 0, 0, 0, 0]
    kids_hist = [0, 0, 0, 0, 0]
    while level:
        nxt = []
        for i in level:
            for c in refs[i]:
                if c == 0xFFFFFFFF:
                    continue
                if c & 0x80000000:
                    k = int(c & 3) + 1
                    leaf_depth_sum += (d + 1) * k; leaf_tris += k; max_leaf_depth = max(max_leaf_depth, d + 1)
                    leaf_sizes[k] += 1
                else:
                    nxt.append(int(c))
        level = nxt
        d += 1
    used = (refs != 0xFFFFFFFF).sum()
    for row in (refs != 0xFFFFFFFF).sum(axis=1):
        kids_hist[int(row)] += 1
    return {"levels": d, "mean_leaf_depth": round(leaf_depth_sum / max(leaf_tris, 1), 2), "max_leaf_depth": max_leaf_depth,
            "children_per_node": round(float(used) / n, 3), "leaves_by_triangle_count": leaf_sizes[1:], "nodes_by_child_count": kids_hist[1:]}


def main():
    out = []
    configs = [a for a in sys.argv[1:] if "=" not in a] or ["config3", "config5"]
    opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    for name in configs:
        mgr = getattr(rtx.scenes, name)(960, 540)
        mgr.numRaysPerPixel = 4
        params, spheres, tris, infos = mgr.build_buffers()
        if "drop_big" in opts:                                # diagnostic: the scene without its few huge triangles (floor, light)
            ext = np.maximum(np.maximum(tris["posA"], tris["posB"]), tris["posC"]) - np.minimum(np.minimum(tris["posA"], tris["posB"]), tris["posC"])
            big = ext.max(axis=1) > 20.0
            keep = np.cumsum(~big) - 1
            infos = infos.copy()
            new_first, new_count = [], []
            for mi in infos:
                a, n = int(mi["firstTriangleIndex"]), int(mi["numTriangles"])
                sel = ~big[a:a + n]
                new_first.append(int(keep[a:a + n][sel][0]) if sel.any() else 0); new_count.append(int(sel.sum()))
            infos["firstTriangleIndex"], infos["numTriangles"] = new_first, new_count
            tris = tris[~big]
        images = {}
        for builder in ("host", "device"):
            tr = rtx.Tracer(0)
            tr.set_option("device_bvh", 1 if builder == "device" else 0)
            tr.set_option("kernel", 1)
            for k, v in opts.items():
                if k != "drop_big":
                    tr.set_option(k, int(v))
            tr.set_params(params)
            tr.set_rows(0, 540)
            times = []
            for rep in range(3):                              # rebuilds of the same scene: the first one pays allocations
                tr.upload(spheres=spheres, triangles=tris, meshinfo=infos)
                t0 = time.perf_counter()
                tr.render(0, 0)                               # n_frames = 0: scene build only
                wall = (time.perf_counter() - t0) * 1e3
                times.append((tr.stats()["lastBvhBuildMs"], wall))
            tr.reset_accum()
            tr.render_counting(0, 1)
            st = tr.stats()
            images[builder] = tr.read_last_frame()
            shape = tree_shape(tr.read_bvh()[0]) if len(tris) <= 200000 else {}
            rec = {"workload": name, "builder": builder, "options": opts, "triangles": int(len(tris)), "bvh_nodes": st["numBvhNodes"], "max_stack": st["bvhMaxStack"], "internal_area": st["bvhInternalArea"],
                   "build_ms": [round(t[0], 3) for t in times], "upload_plus_build_wall_ms": [round(t[1], 2) for t in times],
                   "nodes_per_ray": round(st["nodeVisits"] / st["rays"], 3), "tris_per_ray": round(st["triTests"] / st["rays"], 3),
                   "rays": st["rays"], "shape": shape}
            rec["nodes_plus_tris_per_ray"] = round(rec["nodes_per_ray"] + rec["tris_per_ray"], 3)
            out.append(rec)
            tr.close()
        same = np.array_equal(images["host"].view(np.uint32), images["device"].view(np.uint32))
        out[-1]["same_image_as_host_tree"] = bool(same)
        out[-1]["work_ratio_vs_host"] = round(out[-1]["nodes_plus_tris_per_ray"] / out[-2]["nodes_plus_tris_per_ray"], 4)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
