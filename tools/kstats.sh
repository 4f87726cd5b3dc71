#!/bin/bash
# Per-kernel device time of a command under one build of the library (run on the GPU box):
#   tools/kstats.sh <lib.so | -> <tag> <python script and args ...>     e.g.  tools/kstats.sh - head tools/i1_lab.py variant=0 --angles 24
LIB=$1; TAG=$2; shift 2
[ "$LIB" != "-" ] && export SCARPLET_HIP_LIB=$LIB
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats_$TAG
rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/run.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/trace
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-64s %8s %12s %10s %7s" % ("kernel", "calls", "total ms", "avg_us", "%"))
for r in rows[:12]:
    n = r["Name"].split("(")[0].replace("void ", "")[:62]
    print("%-64s %8s %12.3f %10.2f %7.2f" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
