"""Row-strip image decomposition across the GPUs of one node + the single frame-end gather (RCCL over xGMI; the same
code runs on gloo for CPU tests).

The path shards into independent units: a pixel depends only on (x, y, W, H, frame, scene) — seeds use global pixel
coordinates (RayTracing.shader:360-362) and accumulation is per pixel (Accumulate.shader:43-54) — so each rank traces
rows [row0, row0+nrows) for all frames with no data-path exchange, and one gather to rank 0 collects the strips.
A gather on a fully connected xGMI node is N-1 independent point-to-point transfers into the root.
"""
from __future__ import annotations

from typing import Optional, Tuple


def row_strip(height: int, world: int, rank: int) -> Tuple[int, int, int]:
    """-> (row0, nrows, rows_per_rank).  Contiguous strips of ceil(H/world) rows; trailing ranks may get fewer (or
    zero) rows; rows_per_rank is the padded strip height every rank contributes to the gather."""
    if world < 1 or not (0 <= rank < world) or height < 0:
        raise ValueError(f"bad decomposition: height={height} world={world} rank={rank}")
    per = (height + world - 1) // world
    row0 = min(rank * per, height)
    return row0, max(0, min(per, height - row0)), per


def gather_image(strip, height: int, dist=None, dst: int = 0):
    """strip: torch tensor [rows_per_rank, W, 4] (rows beyond this rank's nrows are padding).  Returns the assembled
    [height, W, 4] image on rank `dst`, None elsewhere.  One collective."""
    import torch
    if dist is None or not dist.is_initialized():
        return strip[:height]
    world, rank = dist.get_world_size(), dist.get_rank()
    bufs = [torch.empty_like(strip) for _ in range(world)] if rank == dst else None
    dist.gather(strip, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat(bufs, dim=0)[:height]


# ---- interleaved 8-row bands: same single gather, balanced load ---------------------------------------------------------
BAND = 8


def band_rows(height: int, world: int, rank: int):
    """Global row indices rendered by `rank` when 8-row bands are dealt round-robin (band b -> rank b % world), in the
    order the rank stores them.  Sky rows and object rows then spread evenly over the ranks."""
    if world < 1 or not (0 <= rank < world) or height < 0:
        raise ValueError(f"bad decomposition: height={height} world={world} rank={rank}")
    rows = []
    for y0 in range(rank * BAND, height, world * BAND):
        rows.extend(range(y0, min(y0 + BAND, height)))
    return rows


def band_rows_padded(height: int, world: int) -> int:
    """Rows every rank contributes to the gather (the largest per-rank row count)."""
    return max((len(band_rows(height, world, r)) for r in range(world)), default=0)


def gather_image_banded(strip, height: int, dist=None, dst: int = 0):
    """strip: torch tensor [band_rows_padded, W, 4] holding this rank's bands back to back.  One gather; rank `dst`
    scatters the gathered rows to their image positions and returns [height, W, 4], the others None."""
    import torch
    if dist is None or not dist.is_initialized():
        return strip[:height]
    world, rank = dist.get_world_size(), dist.get_rank()
    bufs = [torch.empty_like(strip) for _ in range(world)] if rank == dst else None
    dist.gather(strip, bufs, dst=dst)
    if rank != dst:
        return None
    image = torch.empty((height,) + tuple(strip.shape[1:]), dtype=strip.dtype, device=strip.device)
    for r in range(world):
        rows = band_rows(height, world, r)
        if rows:
            image[torch.as_tensor(rows, device=strip.device)] = bufs[r][:len(rows)]
    return image


# ---- the N-rank job's numbers for bench.py's JSON line --------------------------------------------------------------------
def job_report(dist, comm_device, backend: str, rays: float, wall_s: float, kernel_ms: float, gather_ms: float, strip_bytes: int) -> dict:
    """Every rank contributes (rays traced, wall seconds of the timed region, kernel ms, ms spent in the gather call, bytes of its strip);
    returns on EVERY rank: total rays, wall = the slowest rank's, and — for the line's `comm` and `per_rank` objects — what each rank saw.
    `world_seen` is the world size the backend itself reports (what RCCL / gloo initialised with), so a line cannot claim more GPUs than
    took part.  dist = None: a single process."""
    import torch
    if dist is None or not dist.is_initialized():
        return {"total_rays": rays, "wall_s": wall_s, "world_seen": 1,
                "per_rank": {"rays": [int(rays)], "kernel_ms": [round(kernel_ms, 3)], "wall_ms": [round(wall_s * 1e3, 3)]},
                "comm": {"backend": None, "world_seen": 1, "gather_ms": 0.0, "gather_ms_per_rank": [0.0], "bytes": 0}}
    world = dist.get_world_size()
    mine = torch.tensor([float(rays), float(wall_s), float(kernel_ms), float(gather_ms), float(strip_bytes)], dtype=torch.float64, device=comm_device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    rows = [t.tolist() for t in every]
    return {"total_rays": sum(r[0] for r in rows), "wall_s": max(r[1] for r in rows), "world_seen": world,
            "per_rank": {"rays": [int(r[0]) for r in rows], "kernel_ms": [round(r[2], 3) for r in rows], "wall_ms": [round(r[1] * 1e3, 3) for r in rows]},
            "comm": {"backend": backend, "world_seen": world, "gather_ms": round(max(r[3] for r in rows), 3),
                     "gather_ms_per_rank": [round(r[3], 3) for r in rows],
                     "bytes": int(sum(r[4] for r in rows[1:])),              # what travels: every strip but the root's own
                     "note": "one gather to rank 0 at the end of the timed region (N - 1 point-to-point transfers); gather_ms = the slowest rank's "
                             "time inside the collective call, waiting for the other ranks included"}}
