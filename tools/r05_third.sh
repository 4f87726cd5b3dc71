#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
L="--no-cpu-baseline --no-verify --no-e2e --no-other-configs"
run() { cfg=$1; st=$2; shift 2; python bench.py --config $cfg --steps $st --warmup 3 $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$*', d['ms_per_step'], d['config']['tiles'], d['kernels_ms_per_step'], d['gpu'].get('clock_mhz'))"; }
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s -k "round5 or batching or split_row or chunks" > $O/gputest.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gputest.txt
for o in "batch_templ=64" "batch_templ=128" "batch_templ=256" "batch_templ=256,split_i1=0"; do run C1F 6 --opt $o; run C2 8 --opt $o; done | tee $O/batch.txt
for o in "split_i1=1" "split_i1=2" "split_i1=3" "split_i1=4" "split_i1=6" "split_i1=8" "split_fill=8192" "split_fill=16384" "split_i1=4,split_fill=8192"; do run C1F 6 --opt $o; done | tee $O/c1f.txt
run C1F 6 --tile-penalty 512=1.6 | tee -a $O/c1f.txt
run C1F 6 --tile-penalty 512=1.6,1024=1.6 | tee -a $O/c1f.txt
for o in "split_fill=4096" "split_fill=8192" "split_i1=4"; do run C5 20 --opt $o; run C1 40 --opt $o; done | tee $O/small.txt
