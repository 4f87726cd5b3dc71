#!/bin/bash
# An alternative build of the library for an A/B on the GPU box: tools/build_alt.sh <name> [-DFLAG=...]...
# -> scarplet_amd/alt/libscarplet_hip_<name>.so (travels with the snapshot; loaded with SCARPLET_HIP_LIB=<path>)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/alt_$NAME
rm -rf $W && mkdir -p $W/scarplet_amd $W/include $ROOT/scarplet_amd/alt
cp -r $ROOT/scarplet_amd/csrc $W/scarplet_amd/ && cp $ROOT/include/*.h $W/include/
rm -f $W/scarplet_amd/csrc/*.o
sed -i "s|^CXXFLAGS = \(.*\)$|CXXFLAGS = \1 $*|" $W/scarplet_amd/csrc/Makefile
make -s -C $W/scarplet_amd/csrc -j4 ../libscarplet_hip.so
cp $W/scarplet_amd/libscarplet_hip.so $ROOT/scarplet_amd/alt/libscarplet_hip_$NAME.so
echo built scarplet_amd/alt/libscarplet_hip_$NAME.so
