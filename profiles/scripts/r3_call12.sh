#!/bin/bash
# upper bounds: what the PSK slicer's lane-max chain and the equaliser scan cost the carrier wave (stubbed builds, wrong results, timing only)
cd /root/repo
for v in base SLICER EQ BOTH base; do
	if [ $v = base ]; then unset HFDL_GPU_LIB; else export HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_stub_$v.so; fi
	timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'value %.0f steady %.4f demod/blk %.4f pdus %d' % (r['value'], r['steady_state_ms_per_step'], r['demod_kernel_ms_per_block'], r['pdus_in_timed_region']))"
done
