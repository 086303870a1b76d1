#!/bin/bash
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
C=$(python -c "import json; print(json.load(open('/root/repo/profiles/scripts/stamp.json'))['commit'])")
for wl in cfg3 cfg2 cfg4; do
	bash /root/repo/profiles/pmc_passes.sh $wl $OUT $C > /dev/null 2>&1
	cp $OUT/fold_traffic_$wl.json /root/repo/profiles/fold_traffic_$wl.json
done
cd /root/repo
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_cfg3_driver_line.json 2> $OUT/bench2.err
python - <<PY
import json
d = json.load(open("$OUT/bench_cfg3_driver_line.json")); r = d["roofline"]
print("value %.0f frac %.3f traffic %s alg %s ratio %s" % (d["value"], r["frac"], r["traffic"], r["algorithmic_bytes_per_launch"], (r["traffic"] / r["algorithmic_bytes_per_launch"]) if r["traffic"] else None))
print(r["traffic_source"])
PY
