#!/bin/bash
# same-box comparison of builds: bench value and fold-kernel average, two rounds interleaved
cd /root/repo
for round in 1 2; do
for lib in "$@"; do
	HFDL_GPU_LIB=/root/repo/$lib python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
done; done
