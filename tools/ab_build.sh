#!/bin/bash
# usage: tools/ab_build.sh <tag> <source.hip[,source2.hip...]> [extra compiler flags...]  -- an A/B build of the library with the named
# source files compiled with extra flags (e.g. -DGWTF_ENC_DBG=1), linked with the in-tree objects of the others -> build_ab/libgwtf_<tag>.so
# (build_ab/ is git-ignored but travels to the GPU box; load it with tools/bench_train.py --lib ...)
set -e
TAG=$1; SRCS=${2//,/ }; shift 2
cd "$(dirname "$0")/../go_with_the_flows_amd/csrc"
make -s
mkdir -p ../../build_ab
OBJS=""; SKIP=""
for SRC in $SRCS; do
  OBJ=/tmp/ab_${TAG}_${SRC%.hip}.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -Wno-unused-function -Wno-pass-failed "$@" -c $SRC -o $OBJ &
  OBJS="$OBJS $OBJ"; SKIP="$SKIP|^${SRC%.hip}.o\$"
done
wait
OTHERS=$(ls *.o | grep -Ev "${SKIP#|}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/libgwtf_${TAG}.so $OBJS $OTHERS
echo built build_ab/libgwtf_${TAG}.so
