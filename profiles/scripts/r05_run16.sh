#!/bin/bash
# round 5, run 16: the pruned-fold test at the cfg4 size + the fold parity tests after the WIN template change
mkdir -p gpurun_out/r5p
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "pruned or fold" > gpurun_out/r5p/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5p/pytest.log
tail -25 gpurun_out/r5p/pytest.log
