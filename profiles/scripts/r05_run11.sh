#!/bin/bash
# round 5: cfg2 traffic record (halves of 8 blocks) and its bench line
C=$(python -c "import json; print(json.load(open('/root/repo/profiles/scripts/stamp.json'))['commit'])" 2>/dev/null || echo unknown)
OUT=/root/repo/gpurun_out/final2
mkdir -p $OUT
bash /root/repo/profiles/pmc_passes.sh cfg2 $OUT $C > $OUT/pmc_cfg2.log 2>&1
cp $OUT/fold_traffic_cfg2.json /root/repo/profiles/fold_traffic_cfg2.json
cd /root/repo
python bench.py --workload cfg2 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python - <<PY
import json
d = json.load(open("$OUT/bench_cfg2.json")); r = d["roofline"]
print("cfg2", round(d["value"]), round(d["ms_per_step"], 4), "fold", round(r["avg_launch_ms"], 3), r["blocks_per_launch"], "traffic", r["traffic"], r["traffic_source"].get("traffic_over_algorithmic"), "host_ram", d.get("value_host_ram"), "host_path", d["host_path"].get("value"))
PY
tail -3 $OUT/pmc_cfg2.log
