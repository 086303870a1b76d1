#!/bin/bash
# the GPU library's host code under AddressSanitizer through the parity tests (all but the C-program and full-size cases)
cd /root/repo
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
echo "runtime $RT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0 HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_asan.so \
	timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -k "not host_c_program and not full_size and not idle" > gpurun_out/asan.log 2>&1
echo "rc=$?"
tail -5 gpurun_out/asan.log
grep -c "AddressSanitizer" gpurun_out/asan.log
