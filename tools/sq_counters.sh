#!/bin/bash
# SQ counter pass on a short search (own run, no --stats): tools/sq_counters.sh <variant> [time_search args]
cd /tmp && export TMPDIR=/tmp
V=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$V
rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/pmc -- python3 tools/time_search.py --n 10000 --angles 2 --reps 1 --prof 0 --variant $V "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"])[:8]:
    a = acc[k]; wc = a["SQ_WAVE_CYCLES"]
    print("%-42s launches %4d  wave_cycles %.3g  wait_any %.2f  wait_inst %.2f  active_any %.2f  valu %.2f  lds %.2f  bank_conflict %.3f  insts_valu/wave_cycle %.3f" % (
        k, cnt[k], wc, a["SQ_WAIT_ANY"]/wc, a["SQ_WAIT_INST_ANY"]/wc, a["SQ_ACTIVE_INST_ANY"]/wc, a["SQ_ACTIVE_INST_VALU"]/wc, a["SQ_ACTIVE_INST_LDS"]/wc, a["SQ_LDS_BANK_CONFLICT"]/wc, a["SQ_INSTS_VALU"]/wc))
PY
