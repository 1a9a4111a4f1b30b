#!/bin/bash
# A/B of library builds: tools/ab_libs.sh <rounds> <variant> ...   (ab_libs/librt_<variant>.so, selected through RTX_LIB)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=$1; shift
for i in $(seq $rounds); do for v in "$@"; do
  RTX_LIB=$R/ab_libs/librt_$v.so python3 "$R/bench.py" --no-cpu-baseline --no-roofline ${BENCH_ARGS:---kernel 1} 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done; done | tee $R/gpurun_out/ab_libs.txt | sort | awk '{s[$1]+=$2; n[$1]++} END {for (k in s) printf "%s mean %.1f over %d\n", k, s[k]/n[k], n[k]}'
