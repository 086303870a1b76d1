#!/bin/bash
# round 5, first measurement: MFMA lane layout probe, the matrix-pipe fold against the FMA-chain reference, tiling sweep, bench at
# fold batches 8 / 16 and demodulator batches 1 / 2, host-fed legs through the deep staging ring
OUT=/root/repo/gpurun_out/r5a
mkdir -p $OUT
cd /root/repo
./profiles/micro/mfma_layout > $OUT/mfma_layout.txt 2>&1; cat $OUT/mfma_layout.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fold_mfma or fold_batching or channelizer or prefetched or fft_stream or random_call" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
timeout 300 python profiles/fold_variants.py cfg3 3 4,8,16 > $OUT/fold_variants_cfg3.md 2> $OUT/fold_variants_cfg3.err
grep "^|" $OUT/fold_variants_cfg3.md; grep "reference kernel" $OUT/fold_variants_cfg3.md; tail -3 $OUT/fold_variants_cfg3.err
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
for cfg in "16 2" "16 1" "8 2" "8 1"; do
	set -- $cfg
	HFDL_GPU_FOLD_BATCH=$1 HFDL_GPU_DEMOD_BATCH=$2 timeout 300 python bench.py --steps 96 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_nb$1_db$2.json 2> $OUT/bench_cfg3_nb$1_db$2.err
	summ $OUT/bench_cfg3_nb$1_db$2.json "cfg3 fold_batch=$1 demod_batch=$2"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_20.json 2> $OUT/bench_cfg3_20.err
summ $OUT/bench_cfg3_20.json "cfg3 driver-line 20 steps"
for fmt in cf32 cs16; do
	timeout 300 python bench.py --steps 96 --warmup 16 --no-cpu-baseline --no-extra-legs --host-input --sample-format $fmt > $OUT/bench_cfg3_host_$fmt.json 2> $OUT/bench_cfg3_host_$fmt.err
	summ $OUT/bench_cfg3_host_$fmt.json "cfg3 host-input $fmt"
done
timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
summ $OUT/bench_cfg2.json "cfg2"
tail -2 $OUT/*.err | tail -30
