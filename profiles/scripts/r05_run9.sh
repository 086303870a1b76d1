#!/bin/bash
# round 5 experiment: forward FFTs on a stream of their own beside the (now power-bound, not HBM-bound) fold -- laboratory build switch
OUT=/root/repo/gpurun_out/r5i
mkdir -p $OUT
cd /root/repo
summ() {
python - "$1" "$2" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s value %.0f ms/step %.4f steady %s fold_avg %.3f nb %.1f pdus %d/%d demod/blk %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"], d["demod_kernel_ms_per_block"]))
    print("   streams", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("streams", {}).items() if k != "note"})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
export GPU_MAX_HW_QUEUES=8
HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lab.so timeout 300 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/lab_default.json 2> $OUT/lab_default.err
summ $OUT/lab_default.json "lab library, default path, GPU_MAX_HW_QUEUES=8"
HFDL_GPU_FFT_STREAM=1 HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lab.so timeout 300 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/lab_fft_stream.json 2> $OUT/lab_fft_stream.err
summ $OUT/lab_fft_stream.json "lab library, HFDL_GPU_FFT_STREAM=1"
HFDL_GPU_FFT_STREAM=1 HFDL_GPU_DEMOD_BATCH=1 HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lab.so timeout 300 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/lab_fft_stream_db1.json 2> $OUT/lab_fft_stream_db1.err
summ $OUT/lab_fft_stream_db1.json "lab library, HFDL_GPU_FFT_STREAM=1 demod_batch=1"
HFDL_GPU_FFT_STREAM=1 HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lab.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/lab_fft_stream_20.json 2> $OUT/lab_fft_stream_20.err
summ $OUT/lab_fft_stream_20.json "lab library, HFDL_GPU_FFT_STREAM=1, 20 steps"
unset GPU_MAX_HW_QUEUES
timeout 300 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/product.json 2> $OUT/product.err
summ $OUT/product.json "product library"
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch" $f | tail -n 2; done
