#!/bin/bash
# Timing-only ablations (results are wrong while a bit is set).  The bits exist only in a
# -DSC_ABLATE build, made here into scarplet_amd/libscarplet_hip_ablate.so (git-ignored; the
# shipped libscarplet_hip.so has them folded away, sc_internal.h SC_DBGBIT) and selected through
# the developer hook SCARPLET_HIP_LIB; the option "dbg" exists in that build only.
#   k_fwd_rows_curv : 1 no tile norms, 2 no transform, 4 no stores, 8 no loads
#   k_inv_cols      : 16 no template fetch, 32 no transform, 64 no stores   (complex spectra path;
#                     select it for Scarp/Ricker with variant=8)
#   k_inv_cols_sym  : 1 / 2 no coefficient fetch (mirrored / all), 4 no stores, 8 no transform,
#                     16 .. 256 store shapes and flavours (four-column kernels: variant=6)
#   k_inv_rows      : 1 no fetch, 2 no stages, 4 no last stage + scoring, 8 no best-record traffic
#                     (generic row kernel; select it at T = 512..2048 with variant=9)
# The default kernels (k_inv_cols_w8, k_inv_rows_fast) carry no ablation code.
# usage (GPU box): tools/ablate.sh "<option set>" ...     e.g.  tools/ablate.sh "variant=6,dbg=0" "variant=6,dbg=4"
set -e
cd $GRAFT_REPO_ROOT
if [ ! -f scarplet_amd/libscarplet_hip_ablate.so ]; then
  T=$(mktemp -d); mkdir -p $T/scarplet_amd $T/include
  cp -r scarplet_amd/csrc $T/scarplet_amd/; cp include/*.h $T/include/
  rm -f $T/scarplet_amd/csrc/*.o
  make -C $T/scarplet_amd/csrc -j8 ABLATE=1 >/dev/null
  cp $T/scarplet_amd/libscarplet_hip.so scarplet_amd/libscarplet_hip_ablate.so
fi
export SCARPLET_HIP_LIB=$GRAFT_REPO_ROOT/scarplet_amd/libscarplet_hip_ablate.so
python3 tools/i1_lab.py --angles 24 --reset "variant=0,dbg=0" "$@"
