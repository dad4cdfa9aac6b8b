#!/bin/bash
# A/B builds: the library compiled from the working tree with extra compiler flags, as tools/ab/libkt_engine_<name>.so
# (git-ignored, travels to the GPU box with gpurun; select it with KT_ENGINE_LIB=tools/ab/libkt_engine_<name>.so).
#   tools/build_variant.sh noqueue "-DKT_AGG_NO_QUEUE"
set -eu
NAME=$1; FLAGS=${2:-}
REPO=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/kt_var/$NAME
mkdir -p $W/kube_throttler_amd/csrc $W/include $REPO/tools/ab
rm -rf $W/kube_throttler_amd/csrc/* $W/include/*
cp $REPO/kube_throttler_amd/csrc/*.hip $REPO/kube_throttler_amd/csrc/*.h $REPO/kube_throttler_amd/csrc/*.cpp $REPO/kube_throttler_amd/csrc/Makefile $W/kube_throttler_amd/csrc/
cp $REPO/include/*.h $W/include/
make -C $W/kube_throttler_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS" > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
cp $W/kube_throttler_amd/csrc/libkt_engine.so $REPO/tools/ab/libkt_engine_$NAME.so
echo "built tools/ab/libkt_engine_$NAME.so ($FLAGS)"
