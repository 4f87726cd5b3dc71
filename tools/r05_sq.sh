#!/bin/bash
# SQ counters of the shipped library: the C3 plan on two orientations, and bench.py's C1F (the kernels of the small grids)
cd $GRAFT_REPO_ROOT
bash tools/sq_counters.sh 0 > gpurun_out/sq_c3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_c1f; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/pmc -- python3 bench.py --config C1F --steps 2 --warmup 1 --warmup-seconds 0 --no-cpu-baseline --no-verify --no-e2e > $OUT/log.txt 2>&1
python3 - <<PY > gpurun_out/sq_c1f.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:56]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"])[:8]:
    a = acc[k]; wc = a["SQ_WAVE_CYCLES"]
    print("%-58s launches %4d  wave_cycles %.3g  wait_any %.2f  wait_inst %.2f  active_any %.2f  valu %.2f  lds %.2f  bank_conflict %.3f  insts_valu/wave_cycle %.3f" % (
        k, cnt[k], wc, a["SQ_WAIT_ANY"]/wc, a["SQ_WAIT_INST_ANY"]/wc, a["SQ_ACTIVE_INST_ANY"]/wc, a["SQ_ACTIVE_INST_VALU"]/wc, a["SQ_ACTIVE_INST_LDS"]/wc, a["SQ_LDS_BANK_CONFLICT"]/wc, a["SQ_INSTS_VALU"]/wc))
PY
rm -rf $OUT/pmc
cat gpurun_out/sq_c3.txt | tail -9; cat gpurun_out/sq_c1f.txt
