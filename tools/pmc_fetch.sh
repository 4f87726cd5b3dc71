#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of a short search under engine options (run on the GPU box):
#   tools/pmc_fetch.sh <tag> [time_search args, e.g. --n 10000 --angles 2 --opt sib=0]
# Counters in their own passes (no --stats next to --pmc); the program itself follows `--`.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/time_search.py "$@" --reps 1 --prof 0 > $OUT/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 tools/time_search.py "$@" --reps 1 --prof 0 > $OUT/write.log 2>&1
python3 tools/prof_summary.py $OUT | grep -v "^==  kernel stats" > $OUT/summary.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/summary.txt
