#!/bin/bash
# round 6, call 2: the changed / new GPU tests; the thirty-two-column fold tilings alone (laboratory probe) and in the pipeline
OUT=/root/repo/gpurun_out/r6b
mkdir -p $OUT
cd /root/repo
(time timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_constants.py \
	"tests/test_gpu_parity.py::test_fold_mfma_equals_fma_chain" "tests/test_gpu_parity.py::test_fold_batching_changes_nothing" \
	"tests/test_gpu_parity.py::test_full_size_cfg3_geometry" "tests/test_gpu_configs.py::test_full_size_cfg4_burst_dense" \
	"tests/test_gpu_configs.py::test_eight_rank_rehearsal_full_size_one_stream_channel_sharded" --durations=8) > $OUT/pytest_new.log 2>&1
tail -n 16 $OUT/pytest_new.log
HFDL_GPU_FOLD_BATCH=32 FOLD_VARIANTS=0,1,3,4,5,6,7,8 timeout 600 python profiles/fold_variants.py cfg3 3 16,24,32 > $OUT/fold_variants_cfg3_32col.md 2> $OUT/fold_variants_cfg3_32col.err
cat $OUT/fold_variants_cfg3_32col.md | head -40
tail -n 3 $OUT/fold_variants_cfg3_32col.err
summ() {
python - "$1" "$2" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s value %.0f ms/step %.4f steady %s fold_avg %.3f nb %.1f mfma %.3f hbm %.3f demod/blk %s fill_drain %s pdus %d/%d" % (sys.argv[2], d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["mfma"]["frac"], r["hbm"]["frac"], d["demod_kernel_ms_per_block"], d.get("fill_drain_ms"), d["pdus_matching_sent_payload"], d["pdus_in_timed_region"]))
    print("   streams", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("streams", {}).items() if k != "note"})
    print("   shapes", r["launch_shapes"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for fb in 16 32 16 32; do
	HFDL_GPU_FOLD_BATCH=$fb timeout 400 python bench.py --steps 256 --warmup 32 --no-cpu-baseline --no-extra-legs > $OUT/bench256_fb$fb.json 2> $OUT/bench256_fb$fb.err
	summ $OUT/bench256_fb$fb.json "256 steps fold_batch=$fb"
done
for fb in 16 32; do
	HFDL_GPU_FOLD_BATCH=$fb timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench20_fb$fb.json 2> $OUT/bench20_fb$fb.err
	summ $OUT/bench20_fb$fb.json "20 steps fold_batch=$fb"
done
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch" $f | tail -n 2; done
