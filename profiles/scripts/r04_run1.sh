#!/bin/bash
# round 4, first measurement: GPU suite, fold tiling sweep, bench at fold batches 1 / 2 / 4 / 8
OUT=/root/repo/gpurun_out/r4a
mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout 600 python profiles/fold_variants.py cfg3 3 > $OUT/fold_variants_cfg3.md 2> $OUT/fold_variants_cfg3.err
head -50 $OUT/fold_variants_cfg3.md
for nb in 1 2 4 8; do
	HFDL_GPU_FOLD_BATCH=$nb timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_nb$nb.json 2> $OUT/bench_cfg3_nb$nb.err
	python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_cfg3_nb$nb.json"))
    r = d["roofline"]
    print("nb=$nb value %.0f ms/step %.4f steady %.4f fold_avg %.3f frac %.3f pdus %d/%d demod/blk %s" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["frac"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"], d["demod_kernel_ms_per_block"]))
except Exception as e:
    print("nb=$nb failed", e)
PY
done
HFDL_GPU_FOLD_BATCH=4 timeout 600 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg2_nb4.json 2> $OUT/bench_cfg2_nb4.err
HFDL_GPU_FOLD_BATCH=1 timeout 600 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg2_nb1.json 2> $OUT/bench_cfg2_nb1.err
python - <<PY
import json
for nb in (4, 1):
    try:
        d = json.load(open("$OUT/bench_cfg2_nb%d.json" % nb))
        print("cfg2 nb=%d value %.0f ms/step %.4f steady %.4f fold_avg %.4f demod/blk %s pdus %d/%d" % (nb, d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], d["roofline"]["avg_launch_ms"], d["demod_kernel_ms_per_block"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"]))
    except Exception as e:
        print("cfg2 nb", nb, "failed", e)
PY
