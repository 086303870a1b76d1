#!/bin/bash
cd /root/repo
run() { env "$@" python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4), d['pdus_in_timed_region'])"; }
for round in 1 2 3; do
run HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_vC.so
run X=cur
run HFDL_EXP_DEMOD_AFTER_FFT=1
run HFDL_EXP_DEMOD_MARKER=1
done
