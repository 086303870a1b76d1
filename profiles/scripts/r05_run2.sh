#!/bin/bash
# round 5, second measurement: loads kept D rows ahead in the matrix-pipe fold (scheduling barriers), register-resident FFT passes
OUT=/root/repo/gpurun_out/r5b
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fft_forward or fold_mfma or fold_batching or channelizer or raw_ingest or end_to_end_small or nco" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
timeout 300 python profiles/fft_accuracy.py > $OUT/fft_accuracy.txt 2>&1; cat $OUT/fft_accuracy.txt | grep -v "^\["
timeout 300 python profiles/fold_variants.py cfg3 3 4,8,16 > $OUT/fold_variants_cfg3.md 2> $OUT/fold_variants_cfg3.err
grep "^|" $OUT/fold_variants_cfg3.md
summ() {
python - "$1" "$2" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s value %.0f ms/step %.4f steady %s fold_avg %.3f nb %.1f frac %.3f pdus %d/%d demod/blk %s host_ram %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["frac"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"], d["demod_kernel_ms_per_block"], d.get("value_host_ram")))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for cfg in "16 2" "16 1" "8 2"; do
	set -- $cfg
	HFDL_GPU_FOLD_BATCH=$1 HFDL_GPU_DEMOD_BATCH=$2 timeout 300 python bench.py --steps 96 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_nb$1_db$2.json 2> $OUT/bench_cfg3_nb$1_db$2.err
	summ $OUT/bench_cfg3_nb$1_db$2.json "cfg3 fold_batch=$1 demod_batch=$2"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg3 -- python /root/repo/bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_prof.json 2> $OUT/bench_cfg3_prof.err
cd /root/repo
DB=$(find /tmp/prof_cfg3 -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB "cfg3 -- rocprofv3 --kernel-trace --stats -- python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-extra-legs (r05 work in progress)" > $OUT/kernel_stats_cfg3.md 2>$OUT/kernel_stats.err
head -30 $OUT/kernel_stats_cfg3.md
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_20_legs.json 2> $OUT/bench_cfg3_20_legs.err
summ $OUT/bench_cfg3_20_legs.json "cfg3 driver-line 20 steps, extra legs"
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_cfg3_20_legs.json"))
    print("host_ram", d.get("host_ram_input")); print("host_path", d.get("host_path")); print("cfg2", {k: d["cfg2"].get(k) for k in ("value", "value_host_ram", "demod_kernel_ms_per_block", "steady_state_ms_per_step")} if "cfg2" in d else None)
except Exception as e:
    print("legs failed", e)
PY
tail -n 3 $OUT/*.err | tail -40
