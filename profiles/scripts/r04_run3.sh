#!/bin/bash
OUT=/root/repo/gpurun_out/r4l
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_strict.py tests/test_gpu_parity.py -m gpu -x -q -k "strict or demodulator_stage or fold_batching" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
