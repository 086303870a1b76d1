#!/bin/bash
OUT=/root/repo/gpurun_out/r4f
mkdir -p $OUT
cd /root/repo
nproc > $OUT/nproc.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "eight_rank or two_rank" > $OUT/pytest_8rank.log 2>&1; echo "rc=$?" >> $OUT/pytest_8rank.log
tail -15 $OUT/pytest_8rank.log
timeout 900 python profiles/setup_time.py > $OUT/setup_time.json 2> $OUT/setup_time.err
cat $OUT/setup_time.json
