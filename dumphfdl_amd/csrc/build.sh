#!/bin/bash
# Build libhfdl_gpu.so for gfx950 (cross-compiles without a GPU).  Output lands next to the package (in-tree).
set -e
cd "$(dirname "$0")"
OUT=../libhfdl_gpu.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p ../build
$HIPCC $COMMON -c fft_kernels.hip -o ../build/fft_kernels.o &
$HIPCC $COMMON -c fold_kernels.hip -o ../build/fold_kernels.o &
# demodulator: no FMA contraction, so the fp32 recurrences round exactly like the plain-C oracle's
$HIPCC $COMMON -ffp-contract=off -c demod_kernels.hip -o ../build/demod_kernels.o &
$HIPCC $COMMON -x hip -c hfdl_gpu.cpp -o ../build/hfdl_gpu.o &
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT ../build/fft_kernels.o ../build/fold_kernels.o ../build/demod_kernels.o ../build/hfdl_gpu.o
echo "built $(readlink -f $OUT)"
