#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in a device assembly listing (hipcc -S --cuda-device-only):
    tools/kernel_regs.py file.s [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
    def f(k):
        m = re.search(r"\.%s:\s*(\S+)" % k, blk)
        return m.group(1) if m else "?"
    name = subprocess.run(["c++filt", f("name")], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    if flt and flt not in name:
        continue
    agpr = blk.split("\n")[0].strip()
    print("%-62s vgpr %4s agpr %4s sgpr %4s scratch %5s B  lds %6s" % (name[:62], f("vgpr_count"), agpr, f("sgpr_count"),
                                                                   f("private_segment_fixed_size"), f("group_segment_fixed_size")))
