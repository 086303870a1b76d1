#!/bin/bash
# round 5, run 30: demodulator blocks per launch beside the one-wave-per-SIMD fold (the cap of 2 dates from the two-wave K = 1 fold)
mkdir -p gpurun_out/r5ab
run() {
	env $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; s = d['streams']
print('$1 $2', 'value %.0f' % d['value'], {k: round(v['avg_ms'], 3) for k, v in r['launch_shapes'].items()}, 'A %.2f B %.2f D %.2f' % (s['stream_a_ms'], s['stream_b_ms'], s['stream_d_ms']), 'demod %.3f' % s['per_block_ms']['demod'], 'batch', d['demod_blocks_per_launch'], d['pdus_in_timed_region'])"
}
{
run X=0 ""
run HFDL_GPU_DEMOD_BATCH=1 ""
run HFDL_GPU_DEMOD_BATCH=3 ""
run HFDL_GPU_DEMOD_BATCH=4 ""
run X=0 ""
run X=0 "--steps 20 --warmup 5"
run HFDL_GPU_DEMOD_BATCH=1 "--steps 20 --warmup 5"
run HFDL_GPU_DEMOD_BATCH=4 "--steps 20 --warmup 5"
} | tee gpurun_out/r5ab/demod_batch.txt
