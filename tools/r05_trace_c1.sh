#!/bin/bash
# per-dispatch timeline of a few C1 / C5 steps (rocprofv3 kernel trace): where does a small search's time go?
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05o; rm -rf $O; mkdir -p $O; cd $GRAFT_REPO_ROOT
for C in C1 C5; do
rocprofv3 --kernel-trace --output-format csv -d $O/t_$C -- python3 bench.py --config $C --steps 4 --warmup 2 --warmup-seconds 0 --no-cpu-baseline --no-verify --no-e2e > $O/log_$C.txt 2>&1
python3 - $O/t_$C $C > $O/timeline_$C.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
last = rows[-(n // 6):] if sys.argv[2] == "C1" else rows[-80:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
print("%-60s %10s %10s %9s %8s %s" % ("kernel", "start_us", "dur_us", "gap_us", "grid", "wg"))
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")[:58]
    print("%-60s %10.2f %10.2f %9.2f %8s %s" % (nm, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
          "x".join(str(int(r[k])) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")), r["Workgroup_Size_X"]))
    prev_end = e
PY
rm -rf $O/t_$C
done
cat $O/timeline_C1.txt
