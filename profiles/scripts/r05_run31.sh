#!/bin/bash
# round 5, run 31: radix-256 FFT passes with the LDS exchange one component at a time (19 KiB instead of 37 per workgroup: two fit
# beside the demodulator): same bits as before?  faster in the pipeline?  A/B against the build of the commit before, alternating
mkdir -p gpurun_out/r5ac
python - <<'PY' | tee gpurun_out/r5ac/fft_bits.txt
import os, subprocess, sys, json
code = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, "/root/repo")
import dumphfdl_amd as hf
out = {}
for logn in (18, 20, 23):
    rng = np.random.default_rng(logn)
    x = (rng.standard_normal(1 << logn) + 1j * rng.standard_normal(1 << logn)).astype(np.complex64)
    y = hf.fft_forward(x, shifted=bool(logn & 1))
    out[logn] = hashlib.sha256(y.tobytes()).hexdigest()[:16]
print(out)
'''
res = {}
for lib in ("libhfdl_gpu_old.so", "libhfdl_gpu.so"):
    env = dict(os.environ, HFDL_GPU_LIB="/root/repo/dumphfdl_amd/" + lib)
    res[lib] = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    print(lib, res[lib])
print("forward FFT bit-identical to the build before:", res["libhfdl_gpu_old.so"] == res["libhfdl_gpu.so"])
PY
timeout 300 python profiles/fft_accuracy.py 2>/dev/null | tail -4 | tee gpurun_out/r5ac/fft_accuracy.txt
run() {
	HFDL_GPU_LIB=/root/repo/dumphfdl_amd/$1 timeout 300 python bench.py $2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; s = d['streams']
print('$1 $2', 'value %.0f' % d['value'], {k: round(v['avg_ms'], 3) for k, v in r['launch_shapes'].items()}, 'A %.2f B %.2f' % (s['stream_a_ms'], s['stream_b_ms']), 'fft %.4f demod %.3f' % (s['per_block_ms']['fft'], s['per_block_ms']['demod']), d['pdus_in_timed_region'])"
}
{
for i in 1 2; do run libhfdl_gpu_old.so ""; run libhfdl_gpu.so ""; done
run libhfdl_gpu_old.so "--steps 20 --warmup 5"; run libhfdl_gpu.so "--steps 20 --warmup 5"
run libhfdl_gpu_old.so "--workload cfg2"; run libhfdl_gpu.so "--workload cfg2"
} | tee gpurun_out/r5ac/ab.txt
