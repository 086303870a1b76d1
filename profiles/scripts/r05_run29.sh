#!/bin/bash
# round 5, run 29: the four-column form beside the demodulator: at most two waves per SIMD (product), other tilings (laboratory); the
# driver's 20-step line; and the C host path again (two runs per format)
mkdir -p gpurun_out/r5aa
run() {
	env $1 HFDL_GPU_LIB=/root/repo/dumphfdl_amd/$2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$1 $2', '20 steps: value %.0f' % d['value'], {k: round(v['avg_ms'], 3) for k, v in r['launch_shapes'].items()}, 'demod %.3f' % d['streams']['per_block_ms']['demod'])"
}
{
for i in 1 2; do run X=0 libhfdl_gpu.so; done
for t in 3 4 5; do run HFDL_GPU_FOLD_TILE=$t libhfdl_gpu_lab.so; done
run X=0 libhfdl_gpu.so
} | tee gpurun_out/r5aa/small_form_beside_demod.txt
timeout 600 python profiles/fold_variants.py cfg3 3 1,4 2>/dev/null | grep -E "^\| 16x16x4 \| [0-9] \| 1 " | head -12 | tee gpurun_out/r5aa/small_alone.txt
python - > gpurun_out/r5aa/host_path.json 2> gpurun_out/r5aa/host_path.err <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import bench
import dumphfdl_amd as hf
w = bench.WORKLOADS["cfg3"]
g = hf.plan_geometry(4096, 250 / w["fs"])
x, _ = bench.make_input(w, g.input_size, 0, 1)
out = {fmt: [bench.host_path_leg(w, x, bench.channel_plan(w), fmt) for _ in range(3)] for fmt in ("CS16", "CF32")}
print(json.dumps(out))
PY
python -c "
import json
d = json.load(open('gpurun_out/r5aa/host_path.json'))
for k, v in d.items(): print(k, [round(r['value']) for r in v], [r.get('pipeline_drains') for r in v])"
