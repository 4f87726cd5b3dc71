#!/bin/bash
# rocprofv3 passes for one short search (run on the GPU box via gpurun).
# usage: tools/prof_run.sh <tag> [time_search args...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/time_search.py "$@" --prof 0 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/time_search.py "$@" --prof 0 --reps 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 tools/time_search.py "$@" --prof 0 --reps 1 > $OUT/pmc_write.log 2>&1
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
