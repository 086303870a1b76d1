#!/bin/bash
# Same-box A/B of several builds of libhfdl_gpu.so: per-kernel average durations (rocprofv3 kernel trace) and bench value.
# usage: profiles/ab.sh lib1.so lib2.so ...   (paths relative to the repo root)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
	echo "== $lib"
	rm -rf /tmp/ab_trace
	HFDL_GPU_LIB=/root/repo/$lib rocprofv3 --kernel-trace -d /tmp/ab_trace -- python /root/repo/bench.py --no-cpu-baseline --steps 24 --warmup 4 > /tmp/ab.log 2>&1
	DB=$(find /tmp/ab_trace -name "*.db" | head -1)
	python /root/repo/profiles/timeline_rocpd.py $DB 1 | grep -E "fft_pass|ifft|fold_kernel" | cut -d'|' -f2,6
	for i in 1 2; do HFDL_GPU_LIB=/root/repo/$lib python /root/repo/bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"; done
done
