#!/bin/bash
# lab: the row pass reading whole 128-byte lines (a build with -DSC_I2_LAB_FULLLINES, results wrong, timing and bytes valid)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
run() { tag=$1; shift; python bench.py --config C3 --steps 3 --warmup 1 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'), d['gpu'].get('power_w'))"; }
{ run default; SCARPLET_HIP_LIB=$PWD/scarplet_amd/libscarplet_hip_lab.so run full-lines; run default; SCARPLET_HIP_LIB=$PWD/scarplet_amd/libscarplet_hip_lab.so run full-lines; } | tee $O/ab.txt
bash tools/pmc_fetch.sh r05r_default --n 10000 --angles 2 > $O/pmc_default.txt 2>&1
SCARPLET_HIP_LIB=$PWD/scarplet_amd/libscarplet_hip_lab.so bash tools/pmc_fetch.sh r05r_lab --n 10000 --angles 2 > $O/pmc_lab.txt 2>&1
grep -h "k_inv_rows_fast" $O/pmc_default.txt $O/pmc_lab.txt
FUZZ_ONLY=6,32,41,48,55 timeout 600 python tools/fuzz_oracle.py 60 5 > $O/fuzz_detail.txt 2>&1; grep -v Warning $O/fuzz_detail.txt | cut -c1-900
