#!/bin/bash
# kernel-time table of a short search: tools/quick_stats.sh [time_search args]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/quick
rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/time_search.py "$@" --prof 0 > $OUT/log.txt 2>&1
rm -f $OUT/trace/*/*kernel_trace.csv
python3 tools/prof_summary.py $OUT | head -16
