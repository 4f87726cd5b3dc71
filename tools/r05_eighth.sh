#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "float64_scoring" > $O/t1.txt 2>&1; echo "f64 scoring rc=$?"; grep -E "passed|failed|Error|assert" $O/t1.txt | head
python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "c5_grandcanyon_channel_readme or odd_tile" > $O/t2.txt 2>&1; echo "exact rc=$?"; grep -E "exact|fold |passed|failed|Error" $O/t2.txt | head -20
