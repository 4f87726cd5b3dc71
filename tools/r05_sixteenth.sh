#!/bin/bash
# lab: the deferred record update (SC_I2_RAREWIN) in the 512-cell row kernels too (-DSC_I2_RARE512=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05aa; mkdir -p $O
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
run() { tag=$1; cfg=$2; st=$3; shift 3; python bench.py --config $cfg --steps $st --warmup 3 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', '$cfg', d['ms_per_step'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'))"; }
LAB=$PWD/scarplet_amd/libscarplet_hip_lab.so
{ for i in 1 2; do run default C1F 10; SCARPLET_HIP_LIB=$LAB run rare512 C1F 10; run default C5 60; SCARPLET_HIP_LIB=$LAB run rare512 C5 60; run default C1 400; SCARPLET_HIP_LIB=$LAB run rare512 C1 400; done; } | tee $O/ab.txt
SCARPLET_HIP_LIB=$LAB python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "c1f or c5_grandcanyon_channel_five or split_row or round5" 2>&1 | tail -3
