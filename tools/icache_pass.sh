cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_icache; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" $O/avail.txt | head -60 > $O/avail_icache.txt
ARGS="--steps 16 --warmup 16 --no-cpu-baseline --no-roofline --no-latency"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $O/pmc -- python3 $R/bench.py $ARGS > $O/pmc.log 2>&1 || echo "pass failed"
find $O -name "*counter_collection.csv" | head
