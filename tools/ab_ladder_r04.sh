#!/bin/bash
# Round-4 levers on ONE box, interleaved (box-to-box spread is +-3..5 %, larger than some of the steps):
#   cur      the product build
#   nolists  the same library with primary_lists=0 (no camera-ray candidate lists, no focus cache)
#   noskip   built without -mllvm -structurizecfg-skip-uniform-regions=1
#   w5       built with -DRT_STREAM_WAVES=5 (five waves per SIMD, 30 LDS stack entries)
# Build first:  python tools/build_variant.py cur; python tools/build_variant.py noskip --drop=-structurizecfg-skip-uniform-regions=1;
#               python tools/build_variant.py w5 -DRT_STREAM_WAVES=5
# usage: tools/ab_ladder_r04.sh <rounds> [bench args, e.g. --config 5]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=$1; shift
run() { # name lib extra-args...
  local name=$1 lib=$2; shift 2
  RTX_LIB=$R/ab_libs/librt_$lib.so python3 "$R/bench.py" --no-cpu-baseline --no-roofline --no-latency --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])"
}
for i in $(seq $rounds); do
  run cur cur "$@"; run nolists cur --opt primary_lists=0 "$@"; run noskip noskip "$@"; run w5 w5 --opt stream_stack=30 "$@"
done | tee -a $R/gpurun_out/ab_ladder_r04_raw.txt | sort | awk '{s[$1]+=$2; n[$1]++} END {for (k in s) printf "%s mean %.1f Mrays/s over %d\n", k, s[k]/n[k], n[k]}'
