#!/bin/bash
# the near-tie variant of the row pass on the deferred record update: three waves per SIMD (tree) against four (lab4: spills 40 - 144 B)
cd $GRAFT_REPO_ROOT
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
run() { tag=$1; shift; python bench.py --config C3 --steps 2 --warmup 1 --angles 60 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'))"; }
run plain; run near3 --opt near_window=0.00035; SCARPLET_HIP_LIB=$PWD/scarplet_amd/libscarplet_hip_lab4.so run near4 --opt near_window=0.00035; run near3 --opt near_window=0.00035
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -m gpu -k "exact or c5_grandcanyon or odd_tile or random_searches or near_tie" 2>&1 | tail -2
