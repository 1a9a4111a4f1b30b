#!/bin/bash
# one-off PMC passes: tools/pmc_pass.sh <tag> "<counters pass 1>" "<counters pass 2>" ...   (bench args via BENCH_ARGS)
set -eo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
ARGS=${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline --no-roofline}
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_x$i" -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_x$i.log" 2>&1 || { echo "[pmc] pass '$pass' FAILED"; tail -3 "$OUT/pmc_x$i.log"; }
  echo "[pmc] $pass done"
done
