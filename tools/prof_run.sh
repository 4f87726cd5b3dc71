#!/bin/bash
# rocprofv3 passes of the headline benchmark (run on the GPU box via gpurun):
#   tools/prof_run.sh <tag> [bench.py args...]
# Writes gpurun_out/prof_<tag>/{summary.txt, bench.json, traffic.json, csv files}.
# Counters are collected in their own passes (never together with --stats).  The profiled
# runs skip bench.py's host-side legs (--no-verify --no-e2e --no-cpu-baseline): those fork a
# process pool, and under rocprofv3 the GPU is initialised before the program starts.
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LEAN="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py "$@" $LEAN > $OUT/trace.log 2>&1
# (--mode float32: a two-orientation sample is -pi/2 and +pi/2 - ONE template - which tie in every cell: with the near-tie
#  variant on, the sample's row pass would list 1e8 events a launch, nothing like the 181-orientation search's 8 000)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py "$@" $LEAN --angles 2 --mode float32 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 bench.py "$@" $LEAN --angles 2 --mode float32 > $OUT/pmc_write.log 2>&1
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -f $OUT/trace/*/*kernel_trace.csv          # per-dispatch rows: large, the stats file has the summary
cat $OUT/bench.json; cat $OUT/summary.txt; cat $OUT/traffic.json
