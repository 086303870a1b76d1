#!/bin/bash
# TEST-ONLY builds of libhfdl_gpu.so (never shipped, never loaded by default): the demodulator as the one-lane serial loop of
# tests/hostsim/serial_demod.h on the fixed-sequence elementary functions of tests/hostsim/shared_math.h (-DHFDL_DM_STRICT), with the
# shipped pipeline's fast forms switched back on per HFDL_DM_STRICT_FAST (1 sums, 2 AGC, 4 trig, 8 slicer).  Loaded through
# HFDL_GPU_LIB by profiles/strict_study.py and tests/test_gpu_strict.py.  Output: dumphfdl_amd/strict/libhfdl_gpu_strict_<F>.so
set -e
cd "$(dirname "$0")"
mkdir -p ../strict
pids=""
for F in ${@:-0 1 2 4 8 15}; do
	HFDL_OUT=../strict/libhfdl_gpu_strict_$F.so HFDL_BUILD_DIR=../build/strict_$F HFDL_EXTRA_FLAGS="-DHFDL_DM_STRICT -DHFDL_DM_STRICT_FAST=$F" bash build.sh > /dev/null & pids="$pids $!"
done
for p in $pids; do wait $p; done
ls -la ../strict/
