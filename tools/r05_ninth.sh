#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "round5" > $O/gputest.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|Error|assert" $O/gputest.txt | head
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
run() { cfg=$1; st=$2; shift 2; python bench.py --config $cfg --steps $st --warmup 1 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$*', d['ms_per_step'], d['config']['tiles'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'))"; }
{ run C3 3; run C3 3 --opt variant=17; run C3 3; run C2 8; run C1F 6; run C5 20; } | tee $O/ab.txt
