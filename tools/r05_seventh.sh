#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
run() { cfg=$1; st=$2; shift 2; python bench.py --config $cfg --steps $st --warmup 1 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$*', d['ms_per_step'], d['config']['tiles'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'))"; }
{ run C3 3; run C3 3 --opt y_gb=48; run C3 3 --opt y_gb=16; run C3 3; run C3 3 --opt y_gb=48; } | tee $O/ygb.txt
