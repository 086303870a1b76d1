#!/bin/bash
OUT=/root/repo/gpurun_out/r4p
mkdir -p $OUT
cd /root/repo
run() { # decode_stream hwq steps
	GPU_MAX_HW_QUEUES=$2 HFDL_GPU_DECODE_STREAM=$1 timeout 600 python bench.py --steps $3 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/b.json 2> $OUT/b.err
	python - <<PY
import json
try:
    d = json.load(open("$OUT/b.json")); r = d["roofline"]
    print("decode_stream=$1 hwq=$2 steps=$3 value %.0f ms/step %.4f steady %.4f fold_avg %.3f frac %.3f pdus %d/%d demod/blk %.3f" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["frac"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"], d["demod_kernel_ms_per_block"]))
except Exception as e:
    print("failed", e); print(open("$OUT/b.err").read()[-800:])
PY
}
run 0 4 256
run 2 4 256
run 2 8 256
run 0 8 256
run 2 8 20
