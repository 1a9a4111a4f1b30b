#!/bin/bash
# A/B helper: tools/ab.sh "<bench args A>" "<bench args B>" ...  -> one summary line per variant (same process env)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for a in "$@"; do
  python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline $a 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
print('$a', '| Mrays/s', d['value'], 'ms', d['ms_per_step'], '|', r.get('per_ray'), r.get('phase_lane_utilisation'), r.get('phase_wave_execs_per_ray'))"
done
