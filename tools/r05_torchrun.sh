#!/bin/bash
# the driver's N > 1 launch form on a one-GPU box (two ranks on the device, host transport): does every rank exit 0?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --halo host --size 3000 --angles 4 --ages 4 > $O/line.json 2> $O/err.txt
echo "torchrun rc=$?" | tee $O/rc.txt
tail -c 1500 $O/line.json; echo; tail -15 $O/err.txt
# and the one-process form with torch imported beside the library (what a user script that also uses torch does)
python - > $O/both.txt 2>&1 <<'PY'
import torch, numpy as np
import scarplet_amd as sl
from scarplet_amd.WindowedTemplate import Scarp
from scarplet_amd import synthetic
g = synthetic.synthetic_scarp(600)
r = sl.match(g, Scarp, scale=20., age=10., ang_min=-0.2, ang_max=0.2)
print("ok", np.nanmax(r[3]), torch.__version__, torch.cuda.is_available())
PY
echo "torch + library in one process rc=$?" | tee -a $O/rc.txt; tail -5 $O/both.txt
