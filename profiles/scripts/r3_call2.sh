#!/bin/bash
# round 3, second GPU call: demodulator batching (several blocks per launch on the demodulator-bound geometries).
# Whole GPU suite, the default bench line, cfg2 A/B (a launch per block vs batches), cfg2 kernel trace + timeline.
OUT=/root/repo/gpurun_out/r3b
mkdir -p $OUT
cd /root/repo
C=$(cat profiles/scripts/commit.txt 2>/dev/null || echo unknown)
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
for i in 1 2; do
	HFDL_GPU_DEMOD_BATCH=1 timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg2_batch1_$i.json 2>> $OUT/bench.err
	timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg2_batched_$i.json 2>> $OUT/bench.err
done
for b in 2 3; do
	HFDL_GPU_DEMOD_BATCH=$b timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg2_batch$b.json 2>> $OUT/bench.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r3b/bench_cfg2_*.json")):
    try:
        r = json.load(open(f))
        print(f.split("/")[-1], "value %.0f  ms/step %.4f  steady %.4f  demod/blk %.4f  batch %s  pdus %d/%d" % (r["value"], r["ms_per_step"], r["steady_state_ms_per_step"], r["demod_kernel_ms_per_block"], r["demod_blocks_per_launch"], r["pdus_matching_sent_payload"], r["pdus_in_timed_region"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout 900 python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "bench rc=$?"
timeout 600 python bench.py --workload cfg2 > $OUT/bench_cfg2_full.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python /root/repo/bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg2_under_rocprof.json 2>/dev/null
DB=$(find /tmp/kt2 -name "*.db" | head -1)
python /root/repo/profiles/summarize_rocpd.py $DB "cfg2 -- rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs (256 timed blocks + 8 warm-up; commit $C)" > $OUT/cfg2_kernel_stats.md
python /root/repo/profiles/timeline_rocpd.py $DB 2 > $OUT/cfg2_timeline.md
head -12 $OUT/cfg2_kernel_stats.md
head -40 $OUT/cfg2_timeline.md
