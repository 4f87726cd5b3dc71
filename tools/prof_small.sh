#!/bin/bash
# BASELINE configs C1, C2, C5: the bench line and a rocprofv3 kernel-stats summary per config (run on the GPU box):
#   tools/prof_small.sh <tag>     -> gpurun_out/small_<tag>/bench_C*.json, kernel_stats_C*.csv, summary_C*.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/small_$TAG
rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for C in C1 C2 C5 C1F; do
  python3 bench.py --config $C --steps $([ $C = C1F ] && echo 6 || echo 20) --warmup 3 "$@" > $OUT/bench_$C.json 2> $OUT/bench_$C.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$C -- python3 bench.py --config $C --steps $([ $C = C1F ] && echo 6 || echo 20) --warmup 3 --no-cpu-baseline --no-verify --no-e2e > $OUT/trace_$C.log 2>&1
  cp $OUT/trace_$C/*/*kernel_stats.csv $OUT/kernel_stats_$C.csv 2>/dev/null
  python3 - $OUT/kernel_stats_$C.csv $([ $C = C1F ] && echo 9 || echo 23) > $OUT/summary_$C.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
print("%-70s %8s %12s %10s %7s" % ("kernel", "calls", "ms/step", "avg_us", "%"))
for r in rows[:14]:
    n = r["Name"].split("(")[0].replace("void ", "")[:68]
    print("%-70s %8s %12.4f %10.2f %7.2f" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  rm -rf $OUT/trace_$C
  python3 -c "import json;d=json.load(open('$OUT/bench_$C.json'));print('$C', d['ms_per_step'], 'ms', d['value'], d['roofline']['frac'], d.get('verified'))"
  cat $OUT/summary_$C.txt
done
