#!/bin/bash
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q > $OUT/pytest_gpu_configs_at_head.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu_configs_at_head.log
tail -3 $OUT/pytest_gpu_configs_at_head.log
python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench3.err
python bench.py --workload cfg2 > $OUT/bench_cfg2.json 2>> $OUT/bench3.err
python - <<PY
import json
for f in ("bench_cfg3.json", "bench_cfg2.json"):
    d = json.load(open("$OUT/" + f)); r = d["roofline"]
    print(f, "value %.0f frac %.3f traffic %s ratio %s matches %s" % (d["value"], r["frac"], r["traffic"], (r["traffic"] / r["algorithmic_bytes_per_launch"]) if r["traffic"] else None, r["traffic_source"]["csrc_matches_head"]))
PY
