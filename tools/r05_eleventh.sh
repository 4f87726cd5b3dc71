#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
python bench.py --steps 20 --warmup 2 > $O/bench20.json 2> $O/bench20.err; tail -c 400 $O/bench20.json; echo
timeout 900 python tools/fuzz_oracle.py 60 5 > $O/fuzz_oracle.txt 2>&1; echo "fuzz rc=$?"; tail -8 $O/fuzz_oracle.txt
