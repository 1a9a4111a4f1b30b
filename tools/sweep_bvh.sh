#!/bin/bash
# Builder / traversal option sweep on the GPU box: tools/sweep_bvh.sh [bench args] > gpurun_out/sweep_bvh.txt
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
run() { printf "%-50s " "$*"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-latency $BASE "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['per_ray'], r['bvh'])"; }
for BASE in "--config 3" "--config 5"; do
echo "== $BASE"
run
run --opt max_leaf=1
run --opt max_leaf=3
run --opt max_leaf=4
run --opt bvh_bins=16
run --opt bvh_bins=64
run --opt bvh_cost_exp=90
run --opt bvh_cost_exp=110
run --opt bvh_reinsert=2
run --opt full_sort=1
run
done
