#!/bin/bash
# PC sampling of a short C3-shaped search (run on the GPU box): which instructions of the row pass the waves sit on.
#   tools/pcsample.sh <tag> [method: stochastic|host_trap]
# Its own run (no counters, no other trace domain than the kernel trace); bounded by a timeout of its own.
TAG=$1; METHOD=${2:-stochastic}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pcs_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
UNIT=cycles; INT=1048576
if [ "$METHOD" = host_trap ]; then UNIT=time; INT=1000; fi
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INT \
    --kernel-trace --output-format csv -d $OUT/pcs -- python3 tools/time_search.py --n 10000 --angles 2 --reps 1 --prof 0 > $OUT/log.txt 2>&1
echo "rc=$?" >> $OUT/log.txt
ls -la $OUT/pcs/*/ >> $OUT/log.txt 2>&1
tail -5 $OUT/log.txt
for f in $OUT/pcs/*/*pc_sampling*.csv; do echo $f; head -3 $f; wc -l $f; done 2>/dev/null | head -20
