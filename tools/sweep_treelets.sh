#!/bin/bash
# Device builder: treelet passes (bvh_treelets) x top (bvh_top), traced.  tools/sweep_treelets.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
run() { printf "%-64s " "$*"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-latency --steps 20 --warmup 5 $BASE "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['per_ray'], r['phase_wave_execs_per_ray']['node'], r['phase_wave_execs_per_ray']['triangle'], r['bvh'])"; }
BASE="$*"
run
for top in 1024 0 4096; do for t in 0 2 4 5 6 8; do run --opt device_bvh=1 --opt bvh_top=$top --opt bvh_treelets=$t; done; done
run
