#!/bin/bash
# lab: the I1 -> I2 hand-off row-major (a build with -DSC_Y_ROWMAJOR=1: k_inv_cols_w8 + k_inv_rows_fast only, i.e. C3) against rows2 blocks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; mkdir -p $O
L="--no-cpu-baseline --no-e2e --no-other-configs"
run() { tag=$1; shift; python bench.py --config C3 --steps 3 --warmup 1 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'), d['gpu'].get('power_w'), d.get('verified'), d.get('verification',{}).get('cells_off_the_oracle_argmax'))"; }
{ run rows2 --no-verify; SCARPLET_HIP_LIB=$PWD/scarplet_amd/libscarplet_hip_lab.so run row-major; run rows2 --no-verify; SCARPLET_HIP_LIB=$PWD/scarplet_amd/libscarplet_hip_lab.so run row-major --no-verify; } | tee $O/ab.txt
SCARPLET_HIP_LIB=$PWD/scarplet_amd/libscarplet_hip_lab.so bash tools/pmc_fetch.sh r05s_lab --n 10000 --angles 2 > $O/pmc_lab.txt 2>&1
grep -h "k_inv_rows_fast\|k_inv_cols_w8" $O/pmc_lab.txt
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "exact_mode_on_the_real_space or float64_scoring" -s 2>&1 | tail -8
FUZZ_ONLY=6,32,41,48,55 timeout 600 python tools/fuzz_oracle.py 60 5 2>&1 | grep -v arn | tail -8 | cut -c1-600
