#!/bin/bash
# timing labs (results wrong) of k_inv_cols_w8: 1 no store pass, 2 no stages 2 - 3 (products, stage 1, barriers and stores stay), 3 no coefficient fetch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ae; mkdir -p $O
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
run() { tag=$1; shift; python bench.py --config C3 --steps 2 --warmup 1 --angles 60 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'), d['gpu'].get('power_w'))"; }
{ run default; for v in 1 2 3; do SCARPLET_HIP_LIB=$PWD/scarplet_amd/libscarplet_hip_lab$v.so run lab$v; done; run default; } | tee $O/ab.txt
