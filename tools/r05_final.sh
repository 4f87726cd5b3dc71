#!/bin/bash
# Round-5 closing run (on the GPU box): the whole GPU suite, the headline bench with its rocprofv3 passes, the small configs, the oracle fuzz,
# the driver-form bench line (after the PMC passes, so that it carries the traffic of this very library)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
sha256sum scarplet_amd/libscarplet_hip.so > $O/library.txt
python -m pytest tests -m gpu -q > $O/gputest.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/gputest.txt | tail -2
bash tools/prof_run.sh r05 > $O/prof_run.txt 2>&1; grep -c . $O/prof_run.txt
bash tools/prof_small.sh r05 > $O/prof_small.txt 2>&1; grep -E "^C[0-9]" $O/prof_small.txt
timeout 900 python tools/fuzz_oracle.py 150 11 > $O/fuzz_oracle.txt 2>&1; echo "fuzz rc=$?"; grep -v arn $O/fuzz_oracle.txt | tail -5
timeout 900 python tools/fuzz_oracle.py 100 23 > $O/fuzz_oracle2.txt 2>&1; echo "fuzz rc=$?"; grep -v arn $O/fuzz_oracle2.txt | tail -5
python tools/exact_cost.py 2>&1 | grep -v arn > $O/exact_cost.txt; cat $O/exact_cost.txt
cp gpurun_out/prof_r05/traffic.json profiles/traffic.json
python bench.py --steps 20 --warmup 2 > $O/bench20.json 2> $O/bench20.err; tail -c 200 $O/bench20.json; echo
