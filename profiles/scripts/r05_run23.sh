#!/bin/bash
# round 5, run 23: the K = 4 fold (v_mfma_f32_16x16x4_f32, four alias rows per instruction): bit identity, tilings, pipeline
mkdir -p gpurun_out/r5v
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fold_mfma or fold_batching or channelizer_matches or end_to_end_small" > gpurun_out/r5v/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5v/pytest.log
tail -15 gpurun_out/r5v/pytest.log
timeout 600 python profiles/fold_variants.py cfg3 3 1,4,16 > gpurun_out/r5v/fold_variants_k4.md 2> gpurun_out/r5v/err.txt
grep -E "^\| 16x16x4" gpurun_out/r5v/fold_variants_k4.md | head -40; tail -3 gpurun_out/r5v/err.txt
timeout 600 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r5v/bench_256.json 2> gpurun_out/r5v/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r5v/bench_20.json 2>> gpurun_out/r5v/bench.err
python - <<'PY'
import json
for f in ("bench_256", "bench_20"):
    try:
        d = json.load(open("gpurun_out/r5v/%s.json" % f)); r = d["roofline"]
        print(f, round(d["value"]), "ms/step %.4f" % d["ms_per_step"], r["bound"], "frac %.3f" % r["frac"], {k: round(v["avg_ms"], 3) for k, v in r["launch_shapes"].items()}, "hbm frac %.3f" % r["hbm"]["frac"], d["streams"]["stream_a_ms"], d["streams"]["stream_b_ms"], d["streams"]["per_block_ms"], "pdus", d["pdus_in_timed_region"], d["pdus_matching_sent_payload"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 gpurun_out/r5v/bench.err
