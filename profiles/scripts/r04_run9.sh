#!/bin/bash
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fft_stream or fold_batching or random_call" 2>&1 | tail -3
C=$(python -c "import json; print(json.load(open('/root/repo/profiles/scripts/stamp.json'))['commit'])")
for wl in cfg3 cfg2 cfg4; do
	bash /root/repo/profiles/pmc_passes.sh $wl $OUT $C > /dev/null 2>&1
	cp $OUT/fold_traffic_$wl.json /root/repo/profiles/fold_traffic_$wl.json
	python -c "
import json; t = json.load(open('$OUT/fold_traffic_$wl.json')); print('$wl', t['kernel'], {k: v['traffic_over_algorithmic'] for k, v in t['per_shape'].items()})"
done
python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench3.err
python bench.py --workload cfg2 > $OUT/bench_cfg2.json 2>> $OUT/bench3.err
python bench.py --workload cfg4 > $OUT/bench_cfg4.json 2>> $OUT/bench3.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_cfg3_driver_line.json 2>> $OUT/bench3.err
python - <<PY
import json
for f in ("bench_cfg3.json", "bench_cfg2.json", "bench_cfg4.json", "bench_cfg3_driver_line.json"):
    d = json.load(open("$OUT/" + f)); r = d["roofline"]
    print(f, "value %.0f fold %.3f ms frac %.3f traffic ratio %s matches %s" % (d["value"], r["avg_launch_ms"], r["frac"], (r["traffic"] / r["algorithmic_bytes_per_launch"]) if r["traffic"] else None, r["traffic_source"]["csrc_matches_head"]))
PY
