#!/bin/bash
# Option sweep of the Philox mode on the GPU box (one box, back to back): tools/sweep_philox.sh > gpurun_out/sweep_philox.txt
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
run() { printf "%-60s " "$*"; timeout -k 10 200 python bench.py --rng philox --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('latency_ms_single_frame'))"; }
run
run --opt tiles_per_fetch=24
run --opt tiles_per_fetch=32
run --opt fetch_guide_philox=2
run --opt tiles_per_fetch=32 --opt fetch_guide_philox=2
run --shade-threshold 40
run --shade-threshold 56
run --opt node_min=4
run --opt node_min=8
run --opt node_min=10
run
