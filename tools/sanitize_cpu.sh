#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (GPU sanitizers are not available on the pool): the compiled host's scene loader
# and marshal (on the six scenes re-written by save_unity_scene, and on 120 mutated copies), the BVH builder (with and without
# insertion-based optimisation), and the oracle's loop and search tree (fuzz scenes + the 100k-triangle scene).
set -eo pipefail
R=$(cd "$(dirname "$0")/.." && pwd); W=/tmp/rtx_sanitize; mkdir -p $W; cd "$R/tools/sanitize"
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ $SAN -std=c++17 -ffp-contract=off host_harness.cpp ../../ray-tracing-extended_amd/host_cpp/rt_host.cpp ../../ray-tracing-extended_amd/host_cpp/unity_loader.cpp -x c device_stub.c -o $W/host_harness 2>/dev/null
g++ $SAN -std=c++17 bvh_harness.cpp ../../ray-tracing-extended_amd/csrc/bvh.cpp -o $W/bvh_harness
gcc $SAN -std=c11 -ffp-contract=off -fopenmp oracle_harness.c ../../oracle/rt_oracle.c -lm -o $W/oracle_harness
cd "$R"; python3 - <<'PY'
import os, random, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, rtx_pkg
rtx = rtx_pkg.load()
from rtx_amd import unity_scene
import test_gpu_fuzz as fz
W = "/tmp/rtx_sanitize"
names = ["Chess", "Knight", "Suzanne", "Thumbnail", "Balls_Outdoors", "Reflective_Balls"]
for n in names:
    unity_scene.save_unity_scene(unity_scene.load_scene_npz(f"tests/golden/scenes/{n}.npz", 320, 180), f"{W}/{n}.unity")
src = open(f"{W}/Reflective_Balls.unity", "rb").read(); random.seed(1)
for i in range(120):
    b = bytearray(src); mode = i % 4
    if mode == 0: b = b[:random.randrange(len(b))]
    elif mode == 1:
        for _ in range(random.randrange(1, 50)): b[random.randrange(len(b))] = random.randrange(256)
    elif mode == 2:
        a = random.randrange(len(b)); del b[a:random.randrange(a, min(len(b), a + 5000))]
    else:
        a = random.randrange(len(b)); b[a:a] = bytes(random.randrange(32, 127) for _ in range(random.randrange(1, 300)))
    open(f"{W}/mut{i}.unity", "wb").write(b)
def dump(tag, p, s, t, m):
    np.array(p).tofile(f"{W}/orc_p{tag}.bin"); s.tofile(f"{W}/orc_s{tag}.bin"); t.tofile(f"{W}/orc_t{tag}.bin"); m.tofile(f"{W}/orc_m{tag}.bin")
    np.stack([t["posA"], t["posB"], t["posC"]], 1).astype(np.float32).tofile(f"{W}/pos{tag}.bin")
for seed in (0, 4, 16, 21, 33):
    dump(seed, *fz.random_scene(rtx, seed))
dump(99, *rtx.scenes.config3(96, 54).build_buffers())
PY
export ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1
$W/host_harness $W/Chess.unity $W/Knight.unity $W/Suzanne.unity $W/Thumbnail.unity $W/Balls_Outdoors.unity $W/Reflective_Balls.unity
$W/host_harness $W/mut*.unity | grep -c exception | sed 's/^/mutated scenes rejected with an exception: /'
for t in 0 4 16 21 33 99; do $W/bvh_harness $W/pos$t.bin 0; $W/bvh_harness $W/pos$t.bin 2; $W/bvh_harness $W/pos$t.bin 2 1 4; $W/bvh_harness $W/pos$t.bin 0 1 1; $W/bvh_harness $W/pos$t.bin 2 2 4; done
$W/oracle_harness 0 4 16 21 33 99
echo "sanitize_cpu: clean"
