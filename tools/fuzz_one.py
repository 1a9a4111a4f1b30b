#!/usr/bin/env python3
"""One seed of tests/test_gpu_fuzz.py on a fresh context, with option overrides: tools/fuzz_one.py <seed> [name=value ...] [kernel=K]
(bisecting a mismatch the soak run found)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtx_pkg
import oracle_binding
from test_gpu_fuzz import random_scene
from test_gpu_parity import bits_equal, run_gpu

rtx = rtx_pkg.load()
seed = int(sys.argv[1])
over = dict(a.split("=") for a in sys.argv[2:])
b = random_scene(rtx, seed)
kernel = (0, 1, 1, 1, 0, 1, -1)[seed % 7]
rng = np.random.default_rng(seed)
knobs = {"stream_stack": int(rng.choice([4, 9, 30, 37])), "node_min": int(rng.choice([1, 6, 24, 64])), "tiles_per_fetch": int(rng.choice([1, 2, 5, 40])), "fetch_guide": int(rng.choice([1, 4, 16])),
         "max_leaf": int(rng.choice([1, 2, 4])), "full_sort": int(rng.integers(0, 2)), "frame_batch": int(rng.choice([0, 1])),
         "bvh_reinsert": int(rng.choice([0, 0, 2])),
         "stream_tile": int(rng.choice([0, 2, 4])), "compact_nodes": int(rng.choice([0, 1, 1])), "tile_lpt": int(rng.choice([0, 1, 1])),
         "device_bvh": int(rng.choice([0, 1])), "bvh_collapse": int(rng.choice([0, 1, 2])), "bvh_radius": int(rng.choice([2, 8, -16, 40])), "bvh_top": int(rng.choice([0, 2, 600, 1024])), "bvh_treelets": int(rng.choice([0, 1, 3, 6]))}
nf = int(rng.choice([2, 2, 4, 5]))
thr = int(rng.choice([1, 24, 48, 64]))
if "kernel" in over:
    kernel = int(over.pop("kernel"))
if "frames" in over:
    nf = int(over.pop("frames"))
knobs.update({k: int(v) for k, v in over.items()})
tr = rtx.Tracer(0)
for k, v in knobs.items():
    tr.set_option(k, v)
acc, last = run_gpu(tr, b, seed, nf, kernel=kernel, shade_threshold=thr)
st = tr.stats()
want_acc, want_last, cnt = oracle_binding.Oracle().render(*b, seed, nf, accel=True)
bad_last = np.argwhere(~bits_equal(last, want_last).all(-1)); bad_acc = np.argwhere(~bits_equal(acc, want_acc).all(-1))
print(f"seed {seed} kernel {kernel} frames {nf} overrides {over}: last-frame mismatches {len(bad_last)}, accum mismatches {len(bad_acc)}, rays {st['rays']} / {cnt['rays']}, lastKernel {st['lastKernel']}",
      [tuple(x) for x in bad_last[:4]])
tr.close()
