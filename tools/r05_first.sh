#!/bin/bash
# first GPU call of round 5: GPU tests, the MFMA probe, the default bench line, two ranks on the one GPU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
python -m pytest tests -m gpu -x -q -s > $O/gputest.txt 2>&1; echo "gpu tests rc=$?" | tee $O/rc.txt
tail -3 $O/gputest.txt
tools/bin/mfmabench 2000 > $O/mfma.txt 2>&1; cat $O/mfma.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05a/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","verified","gpu","library")})
print(d.get("end_to_end")); print({k:(v.get("ms_per_step"),v.get("roofline_frac"),v.get("verified")) for k,v in d.get("other_configs",{}).items()})
print(d.get("cpu_baseline",{}).get("value"))
PY
timeout 900 python bench.py --gpus 2 --halo host --steps 1 --warmup 0 --no-cpu-baseline > $O/two_ranks.json 2> $O/two_ranks.err; echo "two ranks rc=$?" | tee -a $O/rc.txt
tail -c 1500 $O/two_ranks.json; tail -5 $O/two_ranks.err
