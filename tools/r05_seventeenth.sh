#!/bin/bash
# lab: k_inv_cols_h2's last-stage exchange through ds_swizzle (-DSC_H2_XLANE=2) instead of v_permlane16_swap
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ac; mkdir -p $O
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
run() { tag=$1; cfg=$2; st=$3; shift 3; python bench.py --config $cfg --steps $st --warmup 3 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', '$cfg', d['ms_per_step'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'))"; }
LAB=$PWD/scarplet_amd/libscarplet_hip_lab.so
{ for i in 1 2 3; do run default C1F 10; SCARPLET_HIP_LIB=$LAB run swizzle C1F 10; done; } | tee $O/ab.txt
SCARPLET_HIP_LIB=$LAB python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "c1f or round5 or column_pass_forms" 2>&1 | tail -3
