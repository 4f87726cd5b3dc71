#!/bin/bash
# sample clocks / power while the bench runs: tools/power_trace.sh [bench args]
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline "$@" > gpurun_out/pt_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 80); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|fclk|Power|junction|memory" | sed -E 's/.*: //' | tr '\n' ' '
  echo
  sleep 0.7
done
wait $BP
python -c "
import json
d=json.loads(open('gpurun_out/pt_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"
