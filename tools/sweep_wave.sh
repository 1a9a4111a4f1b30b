#!/bin/bash
# usage: tools/sweep_wave.sh "opt=a,b,c" ["opt2=x,y"]   — one bench line per combination (kernel 3), summary to stdout
out=gpurun_out/sweep_wave.txt; : > $out
IFS='=' read k1 v1 <<< "$1"; IFS='=' read k2 v2 <<< "${2:-_=0}"
for a in ${v1//,/ }; do for b in ${v2//,/ }; do
  extra="--opt $k1=$a"; [ "$k2" != "_" ] && extra="$extra --opt $k2=$b"
  timeout -k 10 120 python bench.py --kernel 3 --no-cpu-baseline $extra ${BENCH_ARGS} > gpurun_out/_s.json 2>/dev/null || { echo "FAILED $k1=$a $k2=$b" >> $out; exit 1; }
  python - "$k1=$a $k2=$b" >> $out <<'PY'
import json,sys
j=json.load(open('gpurun_out/_s.json')); r=j.get('roofline') or {}
print(sys.argv[1], j['value'], j['ms_per_step'], r.get('phase_lane_utilisation'), r.get('phase_wave_execs_per_ray'))
PY
done; done
cat $out
