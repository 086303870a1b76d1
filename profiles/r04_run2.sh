#!/bin/bash
OUT=/root/repo/gpurun_out/r4i
mkdir -p $OUT
cd /root/repo
run() { # fftstream decodestream steps wl hwq
	GPU_MAX_HW_QUEUES=$5 HFDL_GPU_FFT_STREAM=$1 HFDL_GPU_DECODE_STREAM=$2 timeout 600 python bench.py --workload $4 --steps $3 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/b.json 2> $OUT/b.err
	python - <<PY
import json
try:
    d = json.load(open("$OUT/b.json"))
    r = d["roofline"]
    print("$4 hwq=$5 fft_stream=$1 decode_stream=$2 steps=$3 value %.0f ms/step %.4f steady %.4f fold_avg %.3f (%.1f blk) frac %.3f pdus %d/%d demod/blk %.3f" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["frac"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"], d["demod_kernel_ms_per_block"]))
except Exception as e:
    print("failed", e); print(open("$OUT/b.err").read()[-1500:])
PY
}
run 1 0 64 cfg3 8
run 0 0 64 cfg3 8
run 1 1 256 cfg2 8
run 0 1 256 cfg2 8
run 1 0 64 cfg3 16
run 1 1 256 cfg2 16
