#!/bin/bash
# round 5, run 24: A/B on ONE box: the 16x16x1_4B fold (commit a76a519, libhfdl_gpu_old.so) against the 16x16x4 fold, alternating
mkdir -p gpurun_out/r5w
for i in 1 2 3; do
	for lib in libhfdl_gpu_old.so libhfdl_gpu.so; do
		HFDL_GPU_LIB=/root/repo/dumphfdl_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; s = d['streams']
print('$lib', 'value %.0f' % d['value'], 'fold %.3f' % r['avg_launch_ms'], 'A %.2f B %.2f' % (s['stream_a_ms'], s['stream_b_ms']), s['per_block_ms'], d['pdus_in_timed_region'])"
	done
done | tee gpurun_out/r5w/ab_256.txt
for lib in libhfdl_gpu_old.so libhfdl_gpu.so; do
	for i in 1 2; do
		HFDL_GPU_LIB=/root/repo/dumphfdl_amd/$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$lib', '20 steps: value %.0f' % d['value'], {k: round(v['avg_ms'], 3) for k, v in r['launch_shapes'].items()})"
	done
done | tee gpurun_out/r5w/ab_20.txt
