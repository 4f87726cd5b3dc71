#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel stats + PMC passes) into a table."""
import csv, glob, os, sys, collections
root = sys.argv[1]

def find(pat):
    return sorted(glob.glob(os.path.join(root, pat), recursive=True))

def short(n):
    n = n.split("(")[0]
    for a, b in (("void ", ""), ("HIP_vector_type<float, 2u>", "f2")):
        n = n.replace(a, b)
    return n[:60]

print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    print("%-62s %8s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
    for r in rows[:14]:
        print("%-62s %8s %12.3f %12.2f %7.2f" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                              float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
for tag, counters in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"])):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in find(tag + "/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == counters[0]:
                cnt[k] += 1
    if acc:
        print("== %s (per-launch means; FETCH/WRITE_SIZE in KiB as reported, uncorrected) ==" % tag)
        for k in sorted(acc, key=lambda k: -cnt[k])[:12]:
            print("%-62s launches %7d  " % (k, cnt[k]) + "  ".join("%s %.4g" % (c, acc[k][c] / max(cnt[k], 1)) for c in counters))
