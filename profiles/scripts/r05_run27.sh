#!/bin/bash
# round 5, run 27: whole GPU suite + smoke + the full bench line (pruned leg, host legs) on the K = 4 fold
mkdir -p gpurun_out/r5z
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5z/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r5z/pytest_gpu.log
tail -6 gpurun_out/r5z/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r5z/bench_256.json 2> gpurun_out/r5z/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5z/bench_256.json")); r = d["roofline"]
print(round(d["value"]), "ms/step %.4f" % d["ms_per_step"], r["bound"], "frac %.3f" % r["frac"], {k: round(v["avg_ms"], 3) for k, v in r["launch_shapes"].items()}, "hbm frac %.3f" % r["hbm"]["frac"], "host_ram", d.get("value_host_ram"))
print("parity", {k: v for k, v in d["parity"].items() if not isinstance(v, (list, dict))})
print("pruned", json.dumps({k: v for k, v in d["pruned_fold"].items() if k not in ("what", "streams")}))
print("pruned streams", d["pruned_fold"].get("streams"))
print("host_path", d["host_path"].get("value"), "cfg2", d["cfg2"].get("value"))
PY
