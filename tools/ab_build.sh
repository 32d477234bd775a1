#!/bin/bash
# usage: tools/ab_build.sh <tag> <file.hip> '<sed expression>' [...more file/sed pairs]
# A/B build: deepcgp_amd/ab/libdcgp_<tag>.so = the current objects with <file.hip> recompiled after the sed edit.
# Select it at run time with DCGP_LIB=deepcgp_amd/ab/libdcgp_<tag>.so (deepcgp_amd/device.py).  Build the main library first.
set -e
TAG=$1; shift
cd "$(dirname "$0")/../deepcgp_amd/csrc"
mkdir -p ../ab
OBJS=$(ls *.o)
while [ $# -ge 2 ]; do
  F=$1; E=$2; shift; shift
  sed "$E" $F > ab_${TAG}_$F
  X=""; [ "$F" = conv_fused.hip ] && X="-mllvm -disable-machine-licm"   # (the Makefile's per-file flag)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $X -c ab_${TAG}_$F -o ab_${TAG}_${F%.hip}.o
  OBJS=$(echo $OBJS | tr ' ' '\n' | grep -v "^${F%.hip}.o$" | tr '\n' ' ')
  OBJS="$OBJS ab_${TAG}_${F%.hip}.o"
  rm -f ab_${TAG}_$F
done
OBJS=$(echo $OBJS | tr ' ' '\n' | grep -v "^ab_" | tr '\n' ' ')" "$(ls ab_${TAG}_*.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../ab/libdcgp_$TAG.so -ldl
rm -f ab_${TAG}_*.o
echo built deepcgp_amd/ab/libdcgp_$TAG.so
