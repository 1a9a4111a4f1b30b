#!/bin/bash
# One rank's share of the 1080p frame (rank 0 of 8 / of 4) against the scheduling options: tools/sweep_rank8.sh > gpurun_out/sweep_rank8.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
run() { printf "%-56s " "$*"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-latency --no-roofline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for n in 8 4; do
run --as-rank-of $n --steps 16
run --as-rank-of $n --steps 16 --opt fetch_guide=1
run --as-rank-of $n --steps 16 --opt fetch_guide=2
run --as-rank-of $n --steps 16 --opt fetch_guide=8
run --as-rank-of $n --steps 16 --opt fetch_guide=2 --opt tiles_per_fetch=8
run --as-rank-of $n --steps 16 --opt fetch_guide=4 --opt tiles_per_fetch=8
run --as-rank-of $n --steps 16 --opt fetch_guide=1 --opt tiles_per_fetch=8
done
run --as-rank-of 8 --steps 16 --rng philox
run --as-rank-of 8 --steps 5
run --as-rank-of 8 --steps 5 --opt fetch_guide=2
