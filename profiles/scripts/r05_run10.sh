#!/bin/bash
# round 5: PMC passes + traffic records at the final csrc, the bench lines that carry them, the tests the last change touches
C=$(python -c "import json; print(json.load(open('/root/repo/profiles/scripts/stamp.json'))['commit'])" 2>/dev/null || echo unknown)
OUT=/root/repo/gpurun_out/final2
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "prefetched or fold_mfma or fold_batching or cfg2 or collect_without or demodulator_batching or random_call or host_c_program" > $OUT/pytest_subset.log 2>&1; echo "rc=$?" >> $OUT/pytest_subset.log
tail -3 $OUT/pytest_subset.log
cd /tmp && export TMPDIR=/tmp
for wl in cfg3 cfg2 cfg4; do
	bash /root/repo/profiles/pmc_passes.sh $wl $OUT $C > $OUT/pmc_$wl.log 2>&1
	cp $OUT/fold_traffic_$wl.json /root/repo/profiles/fold_traffic_$wl.json
done
cd /root/repo
python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench.err
python bench.py --workload cfg2 > $OUT/bench_cfg2.json 2>> $OUT/bench.err
python bench.py --workload cfg4 > $OUT/bench_cfg4.json 2>> $OUT/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_cfg3_driver_line.json 2>> $OUT/bench.err
python - > $OUT/host_path_cfg2.json 2>> $OUT/bench.err <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import bench
import dumphfdl_amd as hf
w = bench.WORKLOADS["cfg2"]
g = hf.plan_geometry(1024, 250 / w["fs"])
x, _ = bench.make_input(w, g.input_size, 0, 1)
print(json.dumps({"cfg2": {fmt: [bench.host_path_leg(w, x, bench.channel_plan(w), fmt) for _ in range(2)] for fmt in ("CS16", "CF32")}}))
PY
cd /tmp
rm -rf /tmp/kt_cfg2
rocprofv3 --kernel-trace --stats -d /tmp/kt_cfg2 -- python /root/repo/bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg2_under_rocprof.json 2>/dev/null
DB=$(find /tmp/kt_cfg2 -name "*.db" | head -1)
python /root/repo/profiles/summarize_rocpd.py $DB "cfg2 -- rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs (256 timed blocks + 8 warm-up, 8 blocks per fold launch; commit $C)" > $OUT/cfg2_kernel_stats.md
python /root/repo/profiles/timeline_rocpd.py $DB 1 > $OUT/cfg2_timeline.md
python - <<PY
import json
for f in ("bench_cfg3", "bench_cfg2", "bench_cfg4", "bench_cfg3_driver_line"):
    try:
        d = json.load(open("$OUT/%s.json" % f)); r = d["roofline"]
        print(f, round(d["value"]), round(d["ms_per_step"], 4), "fold", round(r["avg_launch_ms"], 3), r["blocks_per_launch"], "traffic", r["traffic"], r["traffic_source"].get("traffic_over_algorithmic") if r["traffic_source"] else None, "host_ram", d.get("value_host_ram"))
    except Exception as e:
        print(f, "ERR", e)
PY
