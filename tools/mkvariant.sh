#!/bin/bash
# usage: tools/mkvariant.sh NAME [-Dflag ...]   -> build/libencoder_NAME.so, prints VGPRs of the main kernel
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
cd $ROOT/ec504_imageencoder_amd/csrc
mkdir -p $ROOT/build
FLAGS="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -fno-slp-vectorize -std=c++17 -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -Rpass-analysis=kernel-resource-usage -c m1v_kernels.hip -o $ROOT/build/m1v_kernels_$NAME.o 2> /tmp/mkvariant_$NAME.log || { grep error /tmp/mkvariant_$NAME.log; exit 1; }
make -s encoder_host.o compat_primitives.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/libencoder_$NAME.so $ROOT/build/m1v_kernels_$NAME.o encoder_host.o compat_primitives.o -lm
echo "$NAME: dense $(grep -A9 "k_encode_denseILi1ELb1" /tmp/mkvariant_$NAME.log | grep -E " VGPRs:| ScratchSize" | sed "s/.*remark: *//; s/ \[.*//" | tr "\n" " ") tiles $(grep -A9 "k_encode_tilesILb1" /tmp/mkvariant_$NAME.log | grep -E " VGPRs:| ScratchSize|Occupancy" | sed "s/.*remark: *//; s/ \[.*//" | tr "\n" " ")"
