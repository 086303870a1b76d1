#!/bin/bash
# round 5, run 26: the K = 4 fold with the product tiling (2, 4, 4) + the four-column form on the same taps: bit identity, sweep, bench
mkdir -p gpurun_out/r5y
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "fold_mfma or fold_batching or channelizer_matches or end_to_end_small or pruned or random_call" > gpurun_out/r5y/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5y/pytest.log
tail -15 gpurun_out/r5y/pytest.log
timeout 600 python profiles/fold_variants.py cfg3 3 1,2,4,16 > gpurun_out/r5y/fold_variants_k4.md 2> gpurun_out/r5y/err.txt
grep -E "^\| 16x16x4" gpurun_out/r5y/fold_variants_k4.md | awk -F'|' '$7+0<=4 || ($3+0==2 && $5+0==4)' | head -60; tail -3 gpurun_out/r5y/err.txt
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r5y/bench_256_$i.json 2> gpurun_out/r5y/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r5y/bench_20_$i.json 2>> gpurun_out/r5y/bench.err
done
python - <<'PY'
import json
for f in ("bench_256_1", "bench_20_1", "bench_256_2", "bench_20_2"):
    try:
        d = json.load(open("gpurun_out/r5y/%s.json" % f)); r = d["roofline"]
        print(f, round(d["value"]), "ms/step %.4f" % d["ms_per_step"], r["bound"], "frac %.3f" % r["frac"], {k: round(v["avg_ms"], 3) for k, v in r["launch_shapes"].items()}, "hbm frac %.3f" % r["hbm"]["frac"], "A %.2f B %.2f" % (d["streams"]["stream_a_ms"], d["streams"]["stream_b_ms"]), "pdus", d["pdus_in_timed_region"], d["pdus_matching_sent_payload"])
    except Exception as e:
        print(f, "ERR", e)
PY
