#!/bin/bash
OUT=/root/repo/gpurun_out/r4t
mkdir -p $OUT
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
