#!/bin/bash
# round 5, run 28: the rest of the GPU suite on the K = 4 fold (run 27 stopped at the low-SNR gate)
mkdir -p gpurun_out/r5z
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_configs.py > gpurun_out/r5z/pytest_gpu_rest.log 2>&1; echo "rc=$?" >> gpurun_out/r5z/pytest_gpu_rest.log
tail -6 gpurun_out/r5z/pytest_gpu_rest.log
