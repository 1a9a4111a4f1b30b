#!/bin/bash
# rocprofv3 passes over the headline bench (run on the GPU box through gpurun).  Counters are collected in their own
# runs, never combined with trace domains other than --kernel-trace/--stats.
# usage: tools/profile.sh <tag> [bench args...]
set -eo pipefail
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
ARGS="--steps 16 --warmup 16 --no-cpu-baseline --no-roofline --no-latency $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" $ARGS > "$OUT/stats.log" 2>&1
echo "[profile] stats done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
            "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
            "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TD_TD_BUSY_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1 || echo "[profile] pass '$pass' failed"
  echo "[profile] pmc $pass done"
done
find "$OUT" -name '*.csv' | head -50
