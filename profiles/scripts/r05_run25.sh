#!/bin/bash
# round 5, run 25: which K = 4 tiling lives best beside the demodulator (registers left on a SIMD decide whether its waves fit)?
mkdir -p gpurun_out/r5x
run() {
	env $1 HFDL_GPU_LIB=/root/repo/dumphfdl_amd/$2 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; s = d['streams']
print('$1 $2', 'value %.0f' % d['value'], 'fold %.3f' % r['avg_launch_ms'], 'A %.2f B %.2f' % (s['stream_a_ms'], s['stream_b_ms']), 'demod %.3f fft %.3f' % (s['per_block_ms']['demod'], s['per_block_ms']['fft']), d['pdus_in_timed_region'])"
}
{
run X=0 libhfdl_gpu_old.so
for t in 9 5 12 4 3; do run HFDL_GPU_FOLD_TILE=$t libhfdl_gpu_lab.so; done
run X=0 libhfdl_gpu_old.so
for t in 9 4; do run HFDL_GPU_FOLD_TILE=$t libhfdl_gpu_lab.so; done
} | tee gpurun_out/r5x/tiles_in_pipeline.txt
