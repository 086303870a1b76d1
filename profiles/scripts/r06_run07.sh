#!/bin/bash
# round 6, call 7: (a) which of the fold's resources slows the demodulator down -- the demodulator alone beside a synthetic neighbour
# that uses one resource at a time (profiles/neighbour_probe.py, profiles/micro/neighbour.hip); (b) the fold's LDS stash with a pitch of
# 64 CG + 1 entries instead of + 4 (writes bank on a 32-dword modulus: + 4 puts parts p and p + 2 on one bank, 4-way): A/B
OUT=/root/repo/gpurun_out/r6g
mkdir -p $OUT
cd /root/repo
timeout 900 python profiles/neighbour_probe.py cfg3 120 2> $OUT/neighbour.err | tee $OUT/neighbour_probe_cfg3.md | cut -c1-200
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
B="python bench.py --no-cpu-baseline --no-extra-legs"
for rep in 1 2; do
for lib in lab lab_xpad1; do
HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_$lib.so timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_${lib}_r$rep.json 2> $OUT/b256_$lib.err; summ $OUT/b256_${lib}_r$rep.json "256 steps $lib"
HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_$lib.so timeout 400 $B --steps 20 --warmup 5 > $OUT/b20_${lib}_r$rep.json 2> $OUT/b20_$lib.err; summ $OUT/b20_${lib}_r$rep.json "20 steps $lib"
done
done
for lib in lab lab_xpad1; do
echo "=== fold tilings alone, $lib"
HFDL_GPU_FOLD_BATCH=32 FOLD_VARIANTS=0,1,2,3,4,5 HFDL_GPU_LAB_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_$lib.so timeout 600 python profiles/fold_variants.py cfg3 3 4,16,32 2> $OUT/fv_$lib.err > $OUT/fold_variants_$lib.md
grep "^| " $OUT/fold_variants_$lib.md | head -n 12 | cut -c1-160
done
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch" $f | tail -n 2; done
