#!/bin/bash
# round 6, call 12: slices of alias rows per fold workgroup (laboratory knob HFDL_GPU_FOLD_SLICES): 4 (the rule of round 1: channels x
# slices >= 1024 workgroups) against 2 and 1 -- longer-lived workgroups (a CU holds ONE at a time: every generation pays its dispatch, its
# first loads and its stores with an idle matrix pipe), a quarter of the partial sums
OUT=/root/repo/gpurun_out/r6l
mkdir -p $OUT
cd /root/repo
summ() {
python - "$1" "$2" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s value %.0f ms/step %.4f steady %s bound %s frac %.3f demod/blk %s x%s fill_drain %.2f parity %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["bound"], r["frac"], d["demod_kernel_ms_per_block"], d["demod_blocks_per_launch"], d.get("fill_drain_ms") or 0, json.dumps(d.get("parity"))[:160]))
    print("   streams", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("streams", {}).get("per_block_ms", {}).items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
export HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lab.so
export HFDL_GPU_LAB_LIB=$HFDL_GPU_LIB
for sl in 4 2 1; do
echo "=== alone, $sl slices"
HFDL_GPU_FOLD_SLICES=$sl HFDL_GPU_FOLD_BATCH=32 FOLD_VARIANTS=0,2,3,5 timeout 600 python profiles/fold_variants.py cfg3 3 4,16,32 2> $OUT/fv_s$sl.err > $OUT/fold_variants_s$sl.md
grep "^| " $OUT/fold_variants_s$sl.md | cut -c1-160
done
B="python bench.py --no-cpu-baseline --no-extra-legs"
for rep in 1 2; do
for sl in 4 2 1; do
HFDL_GPU_FOLD_SLICES=$sl timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_s${sl}_r$rep.json 2> $OUT/b256_s$sl.err; summ $OUT/b256_s${sl}_r$rep.json "256 steps, $sl slices"
done
done
for sl in 4 2 1; do
HFDL_GPU_FOLD_SLICES=$sl timeout 400 $B --steps 20 --warmup 5 > $OUT/b20_s${sl}.json 2> $OUT/b20_s$sl.err; summ $OUT/b20_s${sl}.json "20 steps, $sl slices"
done
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch" $f | tail -n 2; done
