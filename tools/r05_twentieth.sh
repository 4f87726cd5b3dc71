#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05aj; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -m gpu -k "float64 or exact or c5_grandcanyon or odd_tile or random_searches" -s 2>&1 | grep -E "exact|passed|failed|Error|assert|random" | cut -c1-330
python tools/exact_cost.py 2>&1 | grep -v arn | tee $O/exact_cost.txt
timeout 900 python tools/fuzz_oracle.py 150 11 > $O/fuzz_oracle.txt 2>&1; echo "fuzz rc=$?"; grep -v arn $O/fuzz_oracle.txt | tail -5
timeout 900 python tools/fuzz_oracle.py 100 23 > $O/fuzz_oracle2.txt 2>&1; echo "fuzz rc=$?"; grep -v arn $O/fuzz_oracle2.txt | tail -5
