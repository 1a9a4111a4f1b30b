#!/bin/bash
# Device builder: number of treelet passes (bvh_treelets) x host-built top (bvh_top; 0 = none), traced.  tools/sweep_treelets.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
run() { printf "%-64s " "$*"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-latency --steps 20 --warmup 5 $BASE "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['per_ray'], r['phase_wave_execs_per_ray']['node'], r['phase_wave_execs_per_ray']['triangle'], r['bvh'])"; }
BASE="$*"
run
for top in 0 1024 4096; do for t in 0 1 2 3 4 6; do run --opt device_bvh=1 --opt bvh_top=$top --opt bvh_treelets=$t; done; done
run
