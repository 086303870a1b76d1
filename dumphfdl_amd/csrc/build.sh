#!/bin/bash
# Build libhfdl_gpu.so for gfx950 (cross-compiles without a GPU).  Output lands next to the package (in-tree).
#   build.sh          the product library: the tilings the front end can pick, no probes, no A/B switches
#   build.sh lab      libhfdl_gpu_lab.so from the same sources with -DHFDL_LAB: + the tiling sweep, the probes of
#                     include/hfdl_gpu_lab.h and the A/B environment switches (profiles/*.py, bench.py's stream-read probe)
set -e
cd "$(dirname "$0")"
if [ "$1" = "lab" ]; then
	OUT=${HFDL_OUT:-../libhfdl_gpu_lab.so}
	BUILD=${HFDL_BUILD_DIR:-../build/lab}
	HFDL_EXTRA_FLAGS="-DHFDL_LAB ${HFDL_EXTRA_FLAGS:-}"
else
	OUT=${HFDL_OUT:-../libhfdl_gpu.so}          # HFDL_OUT / HFDL_EXTRA_FLAGS: side-by-side builds for A/B measurements
	BUILD=${HFDL_BUILD_DIR:-../build}
fi
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
# -Bsymbolic: calls between the library's own entry points stay inside THIS library when the product and the laboratory build
# are loaded into one process
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $OUT $BUILD/fft_kernels.o $BUILD/fold_kernels.o $BUILD/demod_kernels.o $BUILD/hfdl_gpu.o
echo "built $(readlink -f $OUT)"
