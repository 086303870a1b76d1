#!/bin/bash
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /root/repo
for job in "cfg3 6000" "cfg4 6000" "cfg2 40000"; do
	set -- $job
	timeout 600 python bench.py --workload $1 --steps $2 --warmup 8 --no-cpu-baseline --no-extra-legs > $OUT/soak_$1.json 2> $OUT/soak_$1.err
	python - <<PY
import json
d = json.load(open("$OUT/soak_$1.json")); r = d["roofline"]
print("$1 steps $2: value %.0f Msamples/s, ms/step %.4f, fold %.3f ms x %.1f (frac %.3f, %d launches), PDUs %d, matching sent %d, LPDU walk matching %d" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["frac"], r["launches"], d["pdus_in_timed_region"], d["pdus_matching_sent_payload"], d["pdus_lpdu_walk_matching_sent"]))
PY
done
