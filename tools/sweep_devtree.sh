#!/bin/bash
# Device-built tree against the host tree, traced (config 3 unless bench args say otherwise): tools/sweep_devtree.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
run() { printf "%-58s " "$*"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-latency --steps 20 --warmup 5 $BASE "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['per_ray'], r['phase_wave_execs_per_ray']['node'], r['phase_wave_execs_per_ray']['triangle'], r['bvh'], r['camera_ray_lists']['pixels_no_list'])"; }
BASE="$*"
run
for extra in "$@" ""; do :; done
while read -r line; do run --opt device_bvh=1 $line; done <<'LIST'

--opt bvh_treelets=0
--opt bvh_top=0
--opt bvh_top=256
--opt bvh_top=4096
--opt bvh_top=16384
--opt bvh_top=65536
--opt bvh_radius=16
--opt bvh_radius=32
--opt bvh_radius=-8
--opt bvh_top=4096 --opt bvh_radius=16
--opt bvh_top=2048
--opt bvh_top=512
LIST
run
