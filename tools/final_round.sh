#!/bin/bash
# End-of-round measurements on the GPU box (one gpurun call each part): tools/final_round.sh <part>
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"; O=gpurun_out; mkdir -p $O
case "$1" in
tests)
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/final_pytest.log 2>&1 || { tail -30 $O/final_pytest.log; exit 1; }
  tail -2 $O/final_pytest.log
  python -c "import __graft_entry__ as g; g.smoke()" ;;
profiles)
  tools/profile.sh r04c > $O/prof_r04c.log 2>&1
  tools/profile.sh r04c_philox --rng philox > $O/prof_r04c_philox.log 2>&1
  tools/profile.sh r04c_config5 --config 5 > $O/prof_r04c_config5.log 2>&1
  echo profiles done ;;
profiles2)
  tools/profile.sh r04c_config4 --config 4 > $O/prof_r04c_config4.log 2>&1
  tools/profile.sh r04c_config2 --config 2 > $O/prof_r04c_config2.log 2>&1
  tools/profile.sh r04c_config5_philox --config 5 --rng philox > $O/prof_r04c_config5_philox.log 2>&1
  tools/profile.sh r04c_config2_philox --config 2 --rng philox > $O/prof_r04c_config2_philox.log 2>&1
  echo profiles done ;;
bench)
  timeout -k 10 300 python bench.py > $O/bench_r04_config3.json 2> $O/bench_r04_config3.err
  timeout -k 10 300 python bench.py --rng philox > $O/bench_r04_config3_philox.json 2> $O/bench_r04_config3_philox.err
  timeout -k 10 300 python bench.py --config 5 > $O/bench_r04_config5.json 2> $O/bench_r04_config5.err
  timeout -k 10 300 python bench.py --config 5 --rng philox --no-cpu-baseline > $O/bench_r04_config5_philox.json 2> $O/bench_r04_config5_philox.err
  timeout -k 10 300 python bench.py --config 4 > $O/bench_r04_config4.json 2> $O/bench_r04_config4.err
  timeout -k 10 300 python bench.py --config 2 > $O/bench_r04_config2.json 2> $O/bench_r04_config2.err
  timeout -k 10 300 python bench.py --config 2 --rng philox --no-cpu-baseline > $O/bench_r04_config2_philox.json 2> $O/bench_r04_config2_philox.err
  for n in 2 4 8; do timeout -k 10 200 python bench.py --steps 20 --no-cpu-baseline --no-latency --as-rank-of $n > $O/bench_r04_rank0of$n.json 2>> $O/bench_r04_rank.err; done
  RTX_BENCH_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 8 --no-cpu-baseline 2> $O/bench_r04_gloo2.err | tail -1 > $O/bench_r04_gloo2_one_gpu.json
  python tools/bench_bvh_build.py > $O/bench_bvh_build_r04.json 2> $O/bvh.err
  python tools/bench_geometry.py > $O/bench_geometry_r04.json 2>> $O/bvh.err
  python tools/bsum.py $O/bench_r04_config*.json $O/bench_r04_rank0of*.json $O/bench_r04_gloo2_one_gpu.json ;;
validate)
  timeout -k 10 560 python tools/validate_headline.py 16 > $O/validate_headline_r04.txt 2>&1; tail -2 $O/validate_headline_r04.txt
  timeout -k 10 560 python tools/validate_headline.py 16 philox > $O/validate_headline_r04_philox.txt 2>&1; tail -2 $O/validate_headline_r04_philox.txt ;;
summaries)   # (here, after `profiles` has come back: gpurun_out/prof_r04c* -> profiles/*_summary.{json,md} + profiles/pmc_table.json)
  python tools/summarize_profile.py r04c k_stream && python tools/make_pmc_table.py r04c 3 1920 1080 64 16
  python tools/summarize_profile.py r04c_philox k_stream && python tools/make_pmc_table.py r04c_philox 3 1920 1080 64 16 _philox
  python tools/summarize_profile.py r04c_config5 k_stream && python tools/make_pmc_table.py r04c_config5 5 1920 1080 64 16
  python tools/summarize_profile.py r04c_config4 k_stream && python tools/make_pmc_table.py r04c_config4 4 3840 2160 64 16
  python tools/summarize_profile.py r04c_config2 k_trace && python tools/make_pmc_table.py r04c_config2 2 1920 1080 64 16
  python tools/summarize_profile.py r04c_config5_philox k_stream && python tools/make_pmc_table.py r04c_config5_philox 5 1920 1080 64 16 _philox
  python tools/summarize_profile.py r04c_config2_philox k_stream && python tools/make_pmc_table.py r04c_config2_philox 2 1920 1080 64 16 _philox ;;
sensitivity)
  timeout -k 10 900 python tools/oracle_sensitivity.py --size 256 --spp 1024 --out $O/oracle_sensitivity_r03.json > $O/oracle_sensitivity_r03.log 2>&1; tail -2 $O/oracle_sensitivity_r03.log | cut -c1-200 ;;
esac
