"""Regenerates tests/golden/ from the reference's data files and the oracle.  Runs only in the authoring container
(reads /root/reference, which does not exist on the GPU box).

  scenes/<name>.npz      the six Assets/Scenes/*.unity scenes converted to plain numeric arrays (scene DATA: manager
                         settings, transforms, materials and the serialised MeshSplitter chunks — no reference source)
  pcg_kat.json           PCG known-answer vectors, integer-exact restatement of RayTracing.shader:193-199
  scene_totals.json      chunk / triangle totals serialised by the reference itself (RayTracingManager.cs:156-157)
  *_golden.npz           oracle outputs ("parity unpinned" at float level: they pin the oracle against drift and give the
                         GPU tests fixed expected images)
"""
import glob
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtx_pkg  # noqa: E402

rtx = rtx_pkg.load()
from rtx_amd import unity_scene  # noqa: E402
import oracle_binding  # noqa: E402

REF_SCENES = "/root/reference/Assets/Scenes"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(os.path.join(HERE, "scenes"), exist_ok=True)
    totals = {}
    for p in sorted(glob.glob(os.path.join(REF_SCENES, "*.unity"))):
        name = os.path.splitext(os.path.basename(p))[0].replace(" ", "_")
        m = unity_scene.load_unity_scene(p)
        unity_scene.save_scene_npz(m, os.path.join(HERE, "scenes", name + ".npz"))
        totals[name] = m.serialisedInfo
        print(name, m.serialisedInfo)
    json.dump(totals, open(os.path.join(HERE, "scene_totals.json"), "w"), indent=1)

    # PCG KATs: seed -> first four NextRandom outputs, state afterwards
    def pcg(seed, n=4):
        s, out = seed, []
        for _ in range(n):
            s = (s * 747796405 + 2891336453) & 0xFFFFFFFF
            r = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
            out.append(((r >> 22) ^ r) & 0xFFFFFFFF)
        return out, s
    kat = {str(seed): dict(zip(("outputs", "state"), pcg(seed))) for seed in (0, 1, 719393, 2073599, 2171624, 0xFFFFFFFF)}
    json.dump(kat, open(os.path.join(HERE, "pcg_kat.json"), "w"), indent=1)

    orc = oracle_binding.Oracle()
    # configs[0] in full: 256x256, 4 rays, 3 bounces, frame 0
    b = rtx.scenes.config1().build_buffers()
    acc, last, cnt = orc.render(*b, 0, 1)
    np.savez_compressed(os.path.join(HERE, "config1_golden.npz"), frame0=last, accum0=acc,
                        meta=np.frombuffer(json.dumps(dict(counts=cnt, sha256_frame0=sha(last))).encode(), np.uint8))
    # the reference's own scenes at thumbnail size, two accumulated frames, literal FLAT_CHUNKS semantics
    for name, (w, h, rays, bounces) in {"Chess": (96, 54, 3, 4), "Knight": (64, 36, 4, 3), "Reflective_Balls": (64, 36, 3, 6),
                                        "Balls_Outdoors": (64, 36, 8, 6)}.items():
        m = unity_scene.load_scene_npz(os.path.join(HERE, "scenes", name + ".npz"), w, h)
        m.numRaysPerPixel, m.maxBounceCount = rays, bounces
        b = m.build_buffers()
        acc, last, cnt = orc.render(*b, 0, 2)
        np.savez_compressed(os.path.join(HERE, f"{name}_golden.npz"), accum=acc, last=last,
                            meta=np.frombuffer(json.dumps(dict(width=w, height=h, rays=rays, bounces=bounces, frames=2,
                                                               counts=cnt, sha256_accum=sha(acc))).encode(), np.uint8))
        print(name, cnt)


if __name__ == "__main__":
    main()
