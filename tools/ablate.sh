#!/bin/bash
# Timing-only ablations (results are wrong while a bit is set).  The bits exist only in a
# -DSC_ABLATE build, which this script makes into a scratch copy of the library path and
# removes again; the shipped libscarplet_hip.so has them folded away (sc_internal.h SC_DBGBIT).
#   k_fwd_rows_curv : 1 no tile norms, 2 no transform, 4 no stores, 8 no loads
#   k_inv_cols      : 16 no template fetch, 32 no transform, 64 no stores   (complex spectra path;
#                     select it for Scarp/Ricker with --variant 8)
#   k_inv_rows      : 1 no fetch, 2 no stages, 4 no last stage + scoring, 8 no best-record traffic
#                     (generic row kernel; select it at T = 512..2048 with --variant 9)
# The hot kernels (k_inv_cols_sym, k_inv_rows_fast) carry no ablation code.
# usage (GPU box): tools/ablate.sh <variant> <bits>...
set -e
cd $GRAFT_REPO_ROOT
V=$1; shift
cp scarplet_amd/libscarplet_hip.so /tmp/libscarplet_hip.keep
make -C scarplet_amd/csrc clean >/dev/null
make -C scarplet_amd/csrc -j8 ABLATE=1 >/dev/null
for d in "$@"; do
  echo "SC_DBG=$d"
  SC_DBG=$d python tools/time_search.py --n 10000 --angles 2 --reps 2 --prof 1 --variant $V 2>&1 | grep -E "k_fwd|k_inv"
done
rm -f scarplet_amd/csrc/*.o
cp /tmp/libscarplet_hip.keep scarplet_amd/libscarplet_hip.so
