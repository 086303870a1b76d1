#!/bin/bash
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /root/repo
C=$(python -c "import json; print(json.load(open('/root/repo/profiles/scripts/stamp.json'))['commit'])")
for wl in cfg3 cfg2 cfg4; do
	bash /root/repo/profiles/pmc_passes.sh $wl $OUT $C > /dev/null 2>&1
	cp $OUT/fold_traffic_$wl.json /root/repo/profiles/fold_traffic_$wl.json
done
python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench3.err
python bench.py --workload cfg2 > $OUT/bench_cfg2.json 2>> $OUT/bench3.err
python bench.py --workload cfg4 > $OUT/bench_cfg4.json 2>> $OUT/bench3.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_cfg3_driver_line.json 2>> $OUT/bench3.err
python - <<PY
import json
for f in ("bench_cfg3.json", "bench_cfg2.json", "bench_cfg4.json", "bench_cfg3_driver_line.json"):
    d = json.load(open("$OUT/" + f)); r = d["roofline"]
    print(f, "value %.0f ms/step %.4f steady %.4f fold %.3f ms x %.1f frac %.3f traffic ratio %s matches %s" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["frac"], (r["traffic"] / r["algorithmic_bytes_per_launch"]) if r["traffic"] else None, r["traffic_source"]["csrc_matches_head"]))
    if "cfg2" in d and isinstance(d["cfg2"], dict): print("   cfg2 leg", d["cfg2"].get("value"), d["cfg2"].get("error"))
PY
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
