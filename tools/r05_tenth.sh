#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
run() { cfg=$1; st=$2; shift 2; python bench.py --config $cfg --steps $st --warmup 3 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$st', '$*', d['ms_per_step'], d['warmup_steps_run'], d['config']['tiles'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'), d['gpu'].get('power_w'))"; }
{ run C1 20 --warmup-seconds 0; run C1 400; run C1 2000; run C5 20 --warmup-seconds 0; run C5 60; run C5 200; run C2 8 --warmup-seconds 0; run C2 8; run C1F 6 --warmup-seconds 0; run C1F 6; } | tee $O/ab.txt
