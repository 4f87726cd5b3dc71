#!/bin/bash
# A/B of two builds of the library on the full C3 search (run on the GPU box): tools/ab_lib.sh <alt.so> [i1_lab args]
ALT=$1; shift
for i in 1 2; do
  echo "== default library"; python3 tools/i1_lab.py "variant=0" --steps 2 "$@" | tail -1
  echo "== $ALT"; SCARPLET_HIP_LIB=$ALT python3 tools/i1_lab.py "variant=0" --steps 2 "$@" | tail -1
done
