#!/bin/bash
# round 5, run 12: the opt-in pruned fold -- its test, and the bench line with the pruned_fold leg
mkdir -p gpurun_out/r5l
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "pruned" > gpurun_out/r5l/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5l/pytest.log
tail -30 gpurun_out/r5l/pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r5l/bench_cfg3.json 2> gpurun_out/r5l/bench_cfg3.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5l/bench_cfg3.json"))
print("value", d["value"], "host_ram", d["value_host_ram"])
print(json.dumps(d.get("pruned_fold"), indent=1))
PY
