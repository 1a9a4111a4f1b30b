#!/bin/bash
# Sweep of bench.py over tuning options on the GPU box: tools/sweep_bench.sh <tag> <config> "<opt=val ...>" "<opt=val ...>" ...
# One JSON line per variant in gpurun_out/sweep_<tag>.jsonl plus a one-line digest each on stdout.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; CFG=$2; shift 2
mkdir -p "$R/gpurun_out"
OUT=$R/gpurun_out/sweep_$TAG.jsonl
for v in "$@"; do
  args=""
  for kv in $v; do args="$args --opt $kv"; done
  line=$(python3 "$R/bench.py" --config $CFG --steps ${STEPS:-16} --warmup 1 --no-cpu-baseline ${EXTRA:-} $args 2>/dev/null | tail -1)
  echo "{\"variant\": \"$v\", \"config\": $CFG, \"result\": $line}" >> "$OUT"
  echo "$line" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('cfg$CFG [$v]', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', 'util', r.get('phase_lane_utilisation'), 'execs', r.get('phase_wave_execs_per_ray'), r.get('kernel'))"
done
