#!/bin/bash
# FETCH_SIZE (one counter per pass: FETCH_SIZE and WRITE_SIZE together exceed the hardware, rocprofv3 aborts and hangs) of the column pass per launch under engine option i1_pairs (run on the GPU box): tools/pmc_i1_pairs.sh 1 2 3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_i1p; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for P in "$@"; do
  timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p$P -- python3 bench.py --steps 1 --warmup 0 --angles 2 --no-cpu-baseline --no-verify --no-e2e --no-other-configs --opt i1_pairs=$P > $OUT/p$P.log 2>&1
  python3 - $OUT/p$P $P <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    rows += list(csv.DictReader(open(f)))
acc = {}
for r in rows:
    if "k_inv_cols_w8" not in r["Kernel_Name"]: continue
    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
n = len(acc.get("FETCH_SIZE", []))
print("i1_pairs=%s launches %d  FETCH_SIZE sum %.1f MB (x2 per the guide: %.1f MB)" % (
    sys.argv[2], n, sum(acc.get("FETCH_SIZE", [0])) / 1024, 2 * sum(acc.get("FETCH_SIZE", [0])) / 1024))
PY
done
