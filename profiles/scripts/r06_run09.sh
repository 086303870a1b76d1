#!/bin/bash
# round 6, call 11: the thirty-two-column fold with TWO waves per SIMD held to 208 registers (laboratory tiling 25), so that a demodulator
# wave (80) fits beside them -- alone and in the pipeline, against the product tiling (one wave of 420) and the uncapped two-wave one (256)
OUT=/root/repo/gpurun_out/r6k
mkdir -p $OUT
cd /root/repo
summ() {
python - "$1" "$2" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s value %.0f ms/step %.4f steady %s bound %s frac %.3f demod/blk %s x%s fill_drain %.2f" % (sys.argv[2], d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["bound"], r["frac"], d["demod_kernel_ms_per_block"], d["demod_blocks_per_launch"], d.get("fill_drain_ms") or 0))
    print("   streams", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("streams", {}).get("per_block_ms", {}).items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
export HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lab_xpad1.so
export HFDL_GPU_LAB_LIB=$HFDL_GPU_LIB
echo "=== alone"
HFDL_GPU_FOLD_BATCH=32 FOLD_VARIANTS=3,5,25 timeout 600 python profiles/fold_variants.py cfg3 3 16,32 2> $OUT/fv.err > $OUT/fold_variants.md
grep "^| " $OUT/fold_variants.md | cut -c1-160
B="python bench.py --no-cpu-baseline --no-extra-legs"
for rep in 1 2; do
for tile in -1 25 5; do
HFDL_GPU_FOLD_TILE=$tile timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_t${tile}_r$rep.json 2> $OUT/b256_t$tile.err; summ $OUT/b256_t${tile}_r$rep.json "256 steps tile $tile"
done
done
for tile in -1 25; do
HFDL_GPU_FOLD_TILE=$tile timeout 400 $B --steps 20 --warmup 5 > $OUT/b20_t${tile}.json 2> $OUT/b20_t$tile.err; summ $OUT/b20_t${tile}.json "20 steps tile $tile"
HFDL_GPU_FOLD_TILE=$tile HFDL_GPU_DEMOD_BATCH=2 timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_t${tile}_db2.json 2> $OUT/b256_t${tile}_db2.err; summ $OUT/b256_t${tile}_db2.json "256 steps tile $tile, 2 blocks per demodulator launch"
done
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch" $f | tail -n 2; done
