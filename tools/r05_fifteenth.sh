#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05w; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/gputest.txt | tail -2
python bench.py --steps 20 --warmup 2 > $O/bench20.json 2> $O/bench20.err; tail -c 300 $O/bench20.json; echo
python -c "
import sys; sys.path.insert(0, '.')
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -3
