#!/bin/bash
OUT=/root/repo/gpurun_out/r4e
mkdir -p $OUT
cd /root/repo
run() { # nb tile steps
	HFDL_GPU_FOLD_BATCH=$1 HFDL_GPU_FOLD_TILE=$2 timeout 600 python bench.py --steps $3 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/b.json 2> $OUT/b.err
	python - <<PY
import json
try:
    d = json.load(open("$OUT/b.json"))
    r = d["roofline"]
    print("nb=$1 tile=$2 steps=$3 value %.0f ms/step %.4f steady %.4f fold_avg %.3f (%.1f blk) frac %.3f pdus %d/%d demod/blk %.3f" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["frac"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"], d["demod_kernel_ms_per_block"]))
except Exception as e:
    print("nb=$1 tile=$2 failed", e)
PY
}
run 8 1,1,0,2,10 64
run 8 1,1,0,4,6 64
run 8 1,2,0,4,10 64
run 8 1,1,32,4,1 64
run 8 1,1,0,4,10 64
run 4 1,1,0,4,10 64
run 4 1,1,8,8,0 64
run 4 2,1,0,4,6 64
run 8 1,1,0,4,6 20
run 8 1,1,32,4,1 20
