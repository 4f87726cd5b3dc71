#!/bin/bash
# HBM bytes of sc_settle_exact's kernels on a reduced C3 search (run on the GPU box): tools/pmc_settle.sh <tag> [angles]
# Counters in their own passes (no --stats); bench.py's reduced run prints the settle's counters (pairs, taps) in its line.
TAG=$1; ANG=${2:-4}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_settle_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LEAN="--no-cpu-baseline --no-verify --no-e2e --no-other-configs --steps 1 --warmup 1 --angles $ANG"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $LEAN > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -- python3 bench.py $LEAN > $OUT/write.log 2>&1
python3 - <<EOF
import csv, glob, collections, json
for d, cs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"])):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[-2][-40:] if "anonymous" in r["Kernel_Name"] else r["Kernel_Name"][:40]
            if "k_st_" in r["Kernel_Name"] or "window_f64" in r["Kernel_Name"] or "planes<double" in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == cs[0]: cnt[k] += 1
    for k in sorted(acc, key=lambda k: -acc[k][cs[0]]):
        print(d, "%-42s" % k, {c: "%.4g" % (acc[k][c] / max(cnt[k], 1)) for c in cs}, "launches", cnt[k])
for l in open("$OUT/fetch.log"):
    if l.startswith("{"):
        j = json.loads(l); print("settle:", j.get("settle"), "ms_per_step", j.get("ms_per_step"), j["kernels_ms_per_step"])
EOF
rm -rf $OUT/fetch $OUT/write
