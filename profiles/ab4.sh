#!/bin/bash
cd /tmp && export TMPDIR=/tmp
run() {
	echo "== $*"
	rm -rf /tmp/ab_trace
	env "$@" rocprofv3 --kernel-trace -d /tmp/ab_trace -- python /root/repo/bench.py --no-cpu-baseline --steps 24 --warmup 4 > /tmp/ab.log 2>&1
	DB=$(find /tmp/ab_trace -name "*.db" | head -1)
	python /root/repo/profiles/timeline_rocpd.py $DB 1
}
run HFDL_EXP_DEMOD_AFTER_FFT=1
run HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_vC.so
