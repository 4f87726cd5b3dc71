#!/usr/bin/env python3
"""Per-kernel instruction census of a device assembly listing (hipcc -S --cuda-device-only): loads, stores, scratch
traffic, waits for all outstanding loads, branches.  Many vmcnt(0) waits and branches per load = loads issued a few at
a time inside control flow (k_fwd_rows_curv before round 3: 32 loads, 16 waits).   tools/isa_audit.py file.s [filter]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = re.findall(r"^(_Z\w+):\s*;? *@", txt, re.M) or re.findall(r"^(_Z\w+):", txt, re.M)
print("%-58s %6s %6s %6s %7s %7s %6s %6s" % ("kernel", "lines", "gload", "gstore", "scr_ld", "scr_st", "vm(0)", "branch"))
for n in names:
    m = re.search(r"^%s:.*?^\.Lfunc_end" % re.escape(n), txt, re.M | re.S)      # (a kernel may hold several s_endpgm)
    if not m:
        continue
    body = m.group(0)
    dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    dem = dem.split("(")[0].replace("void ", "")
    if flt and flt not in dem:
        continue
    c = lambda pat: len(re.findall(pat, body))
    print("%-58s %6d %6d %6d %7d %7d %6d %6d" % (dem[:58], body.count("\n"), c(r"\bglobal_load"), c(r"\bglobal_store"),
          c(r"\bscratch_load"), c(r"\bscratch_store"), c(r"s_waitcnt vmcnt\(0\)"), c(r"\bs_cbranch")))
