#!/bin/bash
# timing lab (results wrong): k_inv_cols_w8 without its two workgroup barriers per transform - the upper bound of any scheme that decouples
# a wave's transform from the other waves' store pass
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ad; mkdir -p $O
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
run() { tag=$1; shift; python bench.py --config C3 --steps 3 --warmup 1 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'), d['gpu'].get('power_w'))"; }
LAB=$PWD/scarplet_amd/libscarplet_hip_lab.so
{ run default; SCARPLET_HIP_LIB=$LAB run no-barrier; run default; SCARPLET_HIP_LIB=$LAB run no-barrier; } | tee $O/ab.txt
