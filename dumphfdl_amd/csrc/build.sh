#!/bin/bash
# Build libhfdl_gpu.so for gfx950 (cross-compiles without a GPU).  Output lands next to the package (in-tree).
set -e
cd "$(dirname "$0")"
OUT=${HFDL_OUT:-../libhfdl_gpu.so}          # HFDL_OUT / HFDL_EXTRA_FLAGS: side-by-side builds for A/B measurements
BUILD=${HFDL_BUILD_DIR:-../build}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${HFDL_EXTRA_FLAGS:-}"
mkdir -p $BUILD
pids=""
$HIPCC $COMMON -c fft_kernels.hip -o $BUILD/fft_kernels.o & pids="$pids $!"
$HIPCC $COMMON -c fold_kernels.hip -o $BUILD/fold_kernels.o & pids="$pids $!"
# demodulator: no FMA contraction, so the fp32 recurrences round exactly like the plain-C oracle's
$HIPCC $COMMON -ffp-contract=off -c demod_kernels.hip -o $BUILD/demod_kernels.o & pids="$pids $!"
$HIPCC $COMMON -x hip -c hfdl_gpu.cpp -o $BUILD/hfdl_gpu.o & pids="$pids $!"
for p in $pids; do wait $p; done          # set -e: a failed compile stops the build here instead of linking stale objects
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $BUILD/fft_kernels.o $BUILD/fold_kernels.o $BUILD/demod_kernels.o $BUILD/hfdl_gpu.o
echo "built $(readlink -f $OUT)"
