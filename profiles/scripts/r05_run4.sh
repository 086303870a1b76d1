#!/bin/bash
# round 5, fourth measurement: the fold on v_mfma_f32_16x16x1_4B_f32 (octet-interleaved taps written by the forward FFT itself)
OUT=/root/repo/gpurun_out/r5d
mkdir -p $OUT
cd /root/repo
./profiles/micro/mfma_layout > $OUT/mfma_layout.txt 2>&1; cat $OUT/mfma_layout.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fold_mfma or fold_batching or channelizer or end_to_end_small or raw_ingest or other_sample_rates or six_hundred" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
timeout 300 python profiles/fold_variants.py cfg3 3 4,8,16 > $OUT/fold_variants_cfg3.md 2> $OUT/fold_variants_cfg3.err
grep "^|" $OUT/fold_variants_cfg3.md; tail -2 $OUT/fold_variants_cfg3.err
summ() {
python - "$1" "$2" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s value %.0f ms/step %.4f steady %s fold_avg %.3f nb %.1f frac %.3f pdus %d/%d demod/blk %s host_ram %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["frac"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"], d["demod_kernel_ms_per_block"], d.get("value_host_ram")))
    print("   streams", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("streams", {}).items() if k != "note"})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for cfg in "16 2" "16 1" "8 2"; do
	set -- $cfg
	HFDL_GPU_FOLD_BATCH=$1 HFDL_GPU_DEMOD_BATCH=$2 timeout 300 python bench.py --steps 96 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_nb$1_db$2.json 2> $OUT/bench_cfg3_nb$1_db$2.err
	summ $OUT/bench_cfg3_nb$1_db$2.json "cfg3 fold_batch=$1 demod_batch=$2"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_20.json 2> $OUT/bench_cfg3_20.err
summ $OUT/bench_cfg3_20.json "cfg3 driver-line 20 steps"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg3 -- python /root/repo/bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_prof.json 2> $OUT/bench_cfg3_prof.err
cd /root/repo
DB=$(find /tmp/prof_cfg3 -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB "cfg3 -- rocprofv3 --kernel-trace --stats -- python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-extra-legs (r05 work in progress)" > $OUT/kernel_stats_cfg3.md 2>$OUT/kernel_stats.err
head -16 $OUT/kernel_stats_cfg3.md
python profiles/timeline_rocpd.py $DB > $OUT/timeline_cfg3.md 2>> $OUT/kernel_stats.err; wc -l $OUT/timeline_cfg3.md
for f in $OUT/*.err; do echo "== $f"; grep -v "amdgpu.ids\|UserWarning\|dev = torch\|simple_timer" $f | tail -n 3; done
