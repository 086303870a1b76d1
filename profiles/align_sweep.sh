#!/bin/bash
# EXPERIMENT: fold-kernel time vs placement of the big buffers
cd /root/repo
run() {
	echo "== $*"
	env "$@" HFDL_GPU_DEBUG_ALLOC=1 python bench.py --no-cpu-baseline 2>&1 | grep -E "^alloc|^\{" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('alloc'): print(l.strip())
    else:
        d=json.loads(l); print('value', round(d['value'],1), 'fold_ms', round(d['roofline']['avg_launch_ms'],4))"
}
run X=1
run HFDL_GPU_TAPS_ALIGN_LOG2=21
run HFDL_GPU_TAPS_ALIGN_LOG2=30
run HFDL_GPU_TAPS_ALIGN_LOG2=21 HFDL_GPU_BUF_ALIGN_LOG2=21
run HFDL_GPU_TAPS_ALIGN_LOG2=30 HFDL_GPU_BUF_ALIGN_LOG2=26
run HFDL_GPU_TAPS_ALIGN_LOG2=30 HFDL_GPU_BUF_ALIGN_LOG2=26 HFDL_GPU_BUF_SKEW_KB=4
run HFDL_GPU_TAPS_ALIGN_LOG2=30 HFDL_GPU_BUF_ALIGN_LOG2=26 HFDL_GPU_BUF_SKEW_KB=68
run HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_prev.so
run X=2
