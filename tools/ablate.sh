#!/bin/bash
# Timing-only ablations through SC_DBG (results are wrong while a bit is set).  The bits are
# compiled into the kernels that are NOT on the headline path plus the forward row kernel:
#   k_fwd_rows_curv : 1 no tile norms, 2 no transform, 4 no stores, 8 no loads
#   k_inv_cols      : 16 no template fetch, 32 no transform, 64 no stores   (complex spectra path;
#                     select it for Scarp/Ricker with SC_VARIANT=8)
#   k_inv_rows      : 1 no fetch, 2 no stages, 4 no last stage + scoring, 8 no best-record traffic
#                     (generic row kernel; select it at T = 512..2048 with SC_VARIANT=9)
# The hot kernels (k_inv_cols_sym, k_inv_rows_fast) carry no ablation code: the uniform branches
# alone cost them up to 9 %.
for d in "$@"; do
  echo "SC_DBG=$d"
  SC_DBG=$d python tools/time_search.py --n 10000 --angles 2 --reps 2 --prof 1 2>&1 | grep -E "k_fwd|k_inv"
done
