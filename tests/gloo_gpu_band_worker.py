"""Worker of test_two_gloo_ranks_render_their_bands_on_the_gpu (launched by torch.distributed.run, gloo): each rank owns a
real rt_ctx on GPU 0, renders its interleaved 8-row bands with the HIP kernels and the ranks' strips are gathered — the
N > 1 path of bench.py with the collective on gloo."""
import os, sys
sys.path.insert(0, os.environ["RTX_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RTX_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
import rtx_pkg
rtx = rtx_pkg.load()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, frames = 104, 75, 5
b = rtx.scenes.mesh_test_scene(W, H).build_buffers()
params, spheres, tris, infos = b
tr = rtx.Tracer(0)
tr.set_option("kernel", 1)
tr.set_params(params); tr.upload(spheres=spheres, triangles=tris, meshinfo=infos)
tr.set_bands(rank, world)
tr.render(2, frames)
mine = tr.read_accum()
per = rtx.distributed.band_rows_padded(H, world)
strip = torch.zeros(per, W, 4)
strip[:mine.shape[0]] = torch.from_numpy(mine)
img = rtx.distributed.gather_image_banded(strip, H, dist)
rays = torch.tensor([float(tr.stats()["rays"])], dtype=torch.float64)
dist.all_reduce(rays)
if rank == 0:
    tr.set_rows(0, H)
    tr.reset_accum()
    tr.render(2, frames)
    full = tr.read_accum()
    assert np.array_equal(img.numpy().view(np.uint32), full.view(np.uint32)), "assembled image differs from the undivided render"
    assert int(rays.item()) == tr.stats()["rays"], (rays.item(), tr.stats()["rays"])
    print("GLOO_GPU_BANDS_OK")
tr.close()
dist.destroy_process_group()
