#!/bin/bash
# round 5, run 20: whole GPU suite + smoke + the bench lines after the four-column form and the per-shape roofline
mkdir -p gpurun_out/r5s
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5s/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r5s/pytest_gpu.log
tail -4 gpurun_out/r5s/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5s/bench_20.json 2> gpurun_out/r5s/bench.err
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r5s/bench_256.json 2>> gpurun_out/r5s/bench.err
python - <<'PY'
import json
for f in ("bench_20", "bench_256"):
    d = json.load(open("gpurun_out/r5s/%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"]), "ms/step %.4f" % d["ms_per_step"], r["bound"], "frac %.3f" % r["frac"], r["priced_on"], {k: round(v["avg_ms"], 3) for k, v in r["launch_shapes"].items()}, "traffic", r["traffic"], "hbm frac", r["hbm"]["frac"],
          "host_ram", d.get("value_host_ram"), "pruned", (d.get("pruned_fold") or {}).get("value"), (d.get("pruned_fold") or {}).get("pdus_same_as_full_fold"))
PY
