#!/bin/bash
# second GPU call of round 5: the MFMA probe with one workgroup per CU, the new launch forms (bit identity,
# C1F parity), and A/B timings of each form on C3 / C1F / C5 / C2 / C1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
tools/bin/mfmabench 2000 > $O/mfma.txt 2>&1; cat $O/mfma.txt
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s -k "round5 or c1f or split_row" > $O/gputest.txt 2>&1; echo "gpu tests rc=$?" | tee $O/rc.txt
grep -E "round-5|C1F|passed|failed|Error|assert" $O/gputest.txt | head -30
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
for cfg in C1F C5 C1 C2; do
  for opt in "" "split_i1=0" "split_fill=2048" "variant=17" "split_i1=0,split_fill=2048,variant=17"; do
    st=20; [ $cfg = C1F ] && st=6; [ $cfg = C2 ] && st=8; [ $cfg = C1 ] && st=40
    python bench.py --config $cfg --steps $st --warmup 3 $L ${opt:+--opt $opt} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '${opt:-default}', d['ms_per_step'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'))"
  done
done | tee $O/ab_small.txt
for opt in "" "variant=17" "" "variant=17"; do
  python bench.py --steps 3 --warmup 1 $L ${opt:+--opt $opt} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3', '${opt:-default}', d['ms_per_step'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'))"
done | tee $O/ab_c3.txt
