"""Worker of test_two_rank_gloo_strips_assemble_to_the_full_image (launched by torch.distributed.run, gloo)."""
import os, sys
sys.path.insert(0, os.environ["RTX_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RTX_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
import rtx_pkg, oracle_binding
rtx = rtx_pkg.load(); orc = oracle_binding.Oracle()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
b = rtx.scenes.mesh_test_scene(40, 27).build_buffers()      # 27 rows: uneven split (14 + 13)
H, W = 27, 40
row0, nrows, per = rtx.distributed.row_strip(H, world, rank)
acc, _, _ = orc.render(*b, 0, 2, rect=(0, row0, W, row0 + nrows))
strip = torch.zeros(per, W, 4)
strip[:nrows] = torch.from_numpy(acc)
img = rtx.distributed.gather_image(strip, H, dist)
if rank == 0:
    full, _, _ = orc.render(*b, 0, 2)
    assert img.shape == (H, W, 4)
    assert np.array_equal(img.numpy().view(np.uint32), full.view(np.uint32)), "assembled image differs"
    print("GLOO_STRIPS_OK")
else:
    assert img is None
# ---- interleaved 8-row bands, same scene
rows = rtx.distributed.band_rows(H, world, rank)
per = rtx.distributed.band_rows_padded(H, world)
full2, _, _ = orc.render(*b, 0, 2)           # every rank renders the full image here only to pick its rows from it
strip = torch.zeros(per, W, 4)
for i, y in enumerate(rows):
    acc_row, _, _ = orc.render(*b, 0, 2, rect=(0, y, W, y + 1))
    strip[i] = torch.from_numpy(acc_row[0])
img = rtx.distributed.gather_image_banded(strip, H, dist)
if rank == 0:
    assert np.array_equal(img.numpy().view(np.uint32), full2.view(np.uint32)), "banded image differs"
    print("GLOO_BANDS_OK")
# ---- bench.py's N > 1 numbers: every rank reports, every rank gets the same totals; the world size is the backend's own
rep = rtx.distributed.job_report(dist, "cpu", "gloo", rays=1000.0 * (rank + 1), wall_s=0.5 + 0.25 * rank, kernel_ms=400.0 + rank, gather_ms=2.0 + rank,
                                 strip_bytes=per * W * 16)
assert rep["world_seen"] == world == rep["comm"]["world_seen"] and rep["comm"]["backend"] == "gloo"
assert rep["total_rays"] == sum(1000.0 * (r + 1) for r in range(world)) and rep["wall_s"] == 0.5 + 0.25 * (world - 1)
assert rep["per_rank"]["rays"] == [1000 * (r + 1) for r in range(world)] and len(rep["per_rank"]["kernel_ms"]) == world
assert rep["comm"]["gather_ms"] == 2.0 + (world - 1) and rep["comm"]["bytes"] == (world - 1) * per * W * 16
import json
json.dumps(rep)
if rank == 0:
    print("GLOO_JOB_REPORT_OK")
dist.destroy_process_group()
