#!/bin/bash
# round 5, run 33: the driver's command after the last bench.py refactor (dominant_shape factored out)
mkdir -p gpurun_out/r5ae
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5ae/bench_20.json 2> gpurun_out/r5ae/bench.err; echo "rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5ae/bench_20.json")); r = d["roofline"]
print(round(d["value"]), r["bound"], "%.3f" % r["frac"], r["priced_on"], {k: round(v["avg_ms"], 3) for k, v in r["launch_shapes"].items()}, "traffic", r["traffic"], "cpu", d["cpu_baseline"]["value"], "parity", d["parity"]["pdu_multisets_identical"], "pruned", round(d["pruned_fold"]["value"]))
PY
