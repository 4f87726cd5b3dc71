#!/bin/bash
# A/B of library builds on the C3 step, float32 and exact (run on the GPU box): tools/ab_exact.sh [<alt.so> ...]
for i in 1 2; do
  echo "== in-tree library"; python3 tools/settle_check.py c3only 2>&1 | grep "near-tie search\|exact=False" | tail -3 | cut -c1-80
  for ALT in "$@"; do echo "== $ALT"; SCARPLET_HIP_LIB=$PWD/$ALT python3 tools/settle_check.py c3only 2>&1 | grep "near-tie search\|exact=False" | tail -3 | cut -c1-80; done
done
