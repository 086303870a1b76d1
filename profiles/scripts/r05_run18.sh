#!/bin/bash
# round 5, run 18: the four-column form for launches of at most four blocks: parity tests, the sweep at 1 / 2 / 4 blocks, the bench at 20 and 256 steps
mkdir -p gpurun_out/r5q
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fold or random_call or collect_without" > gpurun_out/r5q/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5q/pytest.log
tail -5 gpurun_out/r5q/pytest.log
timeout 600 python profiles/fold_variants.py cfg3 3 1,2,4 > gpurun_out/r5q/fold_variants_small.md 2> gpurun_out/r5q/fold_variants_small.err
grep -E "octet taps|16x16x1" gpurun_out/r5q/fold_variants_small.md | head -40
for i in 1 2; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r5q/bench_20_$i.json 2> gpurun_out/r5q/bench.err; done
timeout 600 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r5q/bench_256.json 2>> gpurun_out/r5q/bench.err
python - <<'PY'
import json
for f in ("bench_20_1", "bench_20_2", "bench_256"):
    d = json.load(open("gpurun_out/r5q/%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"]), "ms/step %.4f" % d["ms_per_step"], "fold avg %.3f" % r["avg_launch_ms"], r["launch_shapes"], "fill_drain", d["fill_drain_ms"], "pdus", d["pdus_in_timed_region"], d["pdus_matching_sent_payload"])
PY
