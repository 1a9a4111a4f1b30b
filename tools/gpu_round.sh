#!/bin/bash
# One gpurun call: GPU tests, then bench lines.  usage: tools/gpu_round.sh <tag> [pytest -k expression]
set -eo pipefail
TAG=${1:-x}; KEXPR=${2:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > gpurun_out/${TAG}_pytest.log 2>&1 || { tail -30 gpurun_out/${TAG}_pytest.log; exit 1; }
else
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1 || { tail -30 gpurun_out/${TAG}_pytest.log; exit 1; }
fi
tail -3 gpurun_out/${TAG}_pytest.log
