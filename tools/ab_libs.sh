#!/bin/bash
# A/B/C... of several builds of the library on the full C3 search, two rounds on one box (run on the GPU box):
#   tools/ab_libs.sh <alt1.so> [<alt2.so> ...]        (the in-tree library first in every round)
for i in 1 2; do
  echo "== in-tree library"; python3 tools/i1_lab.py "variant=0" --steps 2 | tail -1 | cut -c1-60,230-330
  for ALT in "$@"; do echo "== $ALT"; SCARPLET_HIP_LIB=$ALT python3 tools/i1_lab.py "variant=0" --steps 2 | tail -1 | cut -c1-60,230-330; done
done
