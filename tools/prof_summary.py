#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel stats + PMC passes) into a table."""
import csv, glob, os, sys, collections
root = sys.argv[1]

def find(pat):
    return sorted(glob.glob(os.path.join(root, pat), recursive=True))

def short(n):
    n = n.replace("(anonymous namespace)::", "")
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

# ---- traffic.json: PMC bytes per launch of the two inverse kernels, keyed to THIS build of the
# library (bench.py prints `traffic` only when the hash matches).  FETCH_SIZE is doubled as
# MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; WRITE_SIZE as reported.
import hashlib, json
def kib(tag, counter, pred):
    tot, n = 0.0, 0
    for f in find(tag + "/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and pred(r["Kernel_Name"]):
                tot += float(r["Counter_Value"]); n += 1
    return tot / n if n else None
so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scarplet_amd", "libscarplet_hip.so")
h = hashlib.sha256(open(so, "rb").read()).hexdigest()
# every profiling slot of the library (sc_kernel_name) <- the kernels rocprofv3 lists under it
# (sc_settle_exact's kernels - k_st_*, k_window_f64, k_curv_planes<double> - are one bracket per call in the library's
#  profile and run on fewer pairs in the reduced PMC passes: not in this table; 2 % of the exact step)
SLOTS = (("k_curv", ("k_curv_alpha", "k_curv_planes<float")), ("k_windows", ("k_windows",)), ("k_direct", ("k_direct",)),
         ("k_fwd_rows", ("k_fwd_rows",)), ("k_fwd_cols", ("k_fwd_cols", "k_split_templ")),
         ("k_inv_cols", ("k_inv_cols",)), ("k_inv_rows", ("k_inv_rows",)))
out = {"so_sha256": h, "bytes_per_launch": {}, "detail": {},
       "_note": "bytes per kernel launch, per profiling slot of the library, from rocprofv3 --pmc FETCH_SIZE / "
                "WRITE_SIZE (separate passes, bench.py --angles 2 --mode float32: the plain kernels - the near-tie row pass moves the same planes), FETCH_SIZE doubled per MI355X_MICROARCH.md; "
                "k_inv_cols = mean k_inv_cols_w8 launch (option i1_pairs: two tile pairs, the last of a chunk one; 35 templates, all columns), k_inv_rows = "
                "mean k_inv_rows_fast launch; the forward slots average their curvature and template launches"}
for key, pats in SLOTS:
    pred = lambda n, pats=pats: any(p_ in n for p_ in pats)
    f_, w_ = kib("pmc_fetch", "FETCH_SIZE", pred), kib("pmc_write", "WRITE_SIZE", pred)
    if f_ is not None and w_ is not None:
        out["bytes_per_launch"][key] = int(2 * 1024 * f_ + 1024 * w_)
        out["detail"][key] = {"fetch_bytes_corrected": int(2 * 1024 * f_), "write_bytes": int(1024 * w_)}
json.dump(out, open(os.path.join(root, "traffic.json"), "w"), indent=1)
