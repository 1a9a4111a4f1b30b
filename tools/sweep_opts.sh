#!/bin/bash
# usage: BENCH_ARGS="--config 3 --kernel 1" tools/sweep_opts.sh "opt=a,b,c" ["opt2=x,y"]  — one bench line per combination
out=gpurun_out/sweep_opts.txt; : > $out
IFS='=' read k1 v1 <<< "$1"; IFS='=' read k2 v2 <<< "${2:-_=0}"
for a in ${v1//,/ }; do for b in ${v2//,/ }; do
  extra="--opt $k1=$a"; [ "$k2" != "_" ] && extra="$extra --opt $k2=$b"
  timeout -k 10 120 python bench.py --no-cpu-baseline $extra ${BENCH_ARGS} > gpurun_out/_s.json 2>/dev/null || { echo "FAILED $k1=$a $k2=$b" >> $out; exit 1; }
  python - "$k1=$a $k2=$b" >> $out <<'PY'
import json,sys
j=json.load(open('gpurun_out/_s.json')); r=j.get('roofline') or {}
print(sys.argv[1], j['value'], j['ms_per_step'], r.get('per_ray'), (r.get('bvh') or {}).get('nodes'), r.get('phase_wave_execs_per_ray'))
PY
done; done
cat $out
