#!/bin/bash
# TEST-ONLY builds of libhfdl_gpu.so (never shipped, never loaded by default): the demodulator as the one-lane serial loop of
# tests/hostsim/serial_demod.h on the fixed-sequence elementary functions of tests/hostsim/shared_math.h (-DHFDL_DM_STRICT), with the
# shipped pipeline's fast forms switched back on per HFDL_DM_STRICT_FAST (1 sums, 2 AGC, 4 trig, 8 slicer).  Loaded through
# HFDL_GPU_LIB by profiles/strict_study.py and tests/test_gpu_strict.py (which builds the two it needs when they are missing).
# Output: build/strict/libhfdl_gpu_strict_<F>.so at the repository root -- outside the package.  Only the demodulator kernels are
# compiled again (tests/hostsim/strict_demod_kernels.hip wraps demod_kernels.hip); the other objects are the product build's (dumphfdl_amd/build, made by build.sh).
set -e
cd "$(dirname "$0")"
OUTDIR=../../build/strict
mkdir -p $OUTDIR
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for o in fft_kernels.o fold_kernels.o hfdl_gpu.o; do [ -f ../build/$o ] || bash build.sh > /dev/null; done
pids=""
for F in ${@:-0 1 2 4 8 15}; do
	( $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DHFDL_DM_STRICT_FAST=$F -I. -c ../../tests/hostsim/strict_demod_kernels.hip -o $OUTDIR/demod_kernels_$F.o &&
	  $HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $OUTDIR/libhfdl_gpu_strict_$F.so ../build/fft_kernels.o ../build/fold_kernels.o $OUTDIR/demod_kernels_$F.o ../build/hfdl_gpu.o ) & pids="$pids $!"
done
for p in $pids; do wait $p; done
ls -la $OUTDIR/*.so
