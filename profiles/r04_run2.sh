#!/bin/bash
# round 4: fold tiling sweeps + bench at fold batches 4 / 8
OUT=/root/repo/gpurun_out/r4c
mkdir -p $OUT
cd /root/repo
timeout 900 python profiles/fold_variants.py cfg3 3 > $OUT/fold_variants_cfg3.md 2> $OUT/fold_variants_cfg3.err
grep "^|" $OUT/fold_variants_cfg3.md
timeout 900 python profiles/fold_variants.py cfg2 5 > $OUT/fold_variants_cfg2.md 2> $OUT/fold_variants_cfg2.err
grep "^|" $OUT/fold_variants_cfg2.md
for nb in 4 8; do for steps in 64 20; do
	HFDL_GPU_FOLD_BATCH=$nb timeout 600 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_nb${nb}_$steps.json 2> $OUT/bench_cfg3_nb${nb}_$steps.err
	python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_cfg3_nb${nb}_$steps.json"))
    r = d["roofline"]
    print("nb=$nb steps=$steps value %.0f ms/step %.4f steady %.4f fold_avg %.3f (%.1f blk) frac %.3f pdus %d/%d demod/blk %s" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["frac"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"], d["demod_kernel_ms_per_block"]))
except Exception as e:
    print("nb=$nb failed", e)
PY
done; done
for nb in 4 8; do
HFDL_GPU_FOLD_BATCH=$nb timeout 600 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg2_nb$nb.json 2> $OUT/bench_cfg2_nb$nb.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_cfg2_nb$nb.json"))
    print("cfg2 nb=$nb value %.0f ms/step %.4f steady %.4f fold_avg %.4f demod/blk %s pdus %d/%d" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], d["roofline"]["avg_launch_ms"], d["demod_kernel_ms_per_block"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"]))
except Exception as e:
    print("cfg2 failed", e)
PY
done
