#!/bin/bash
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /root/repo
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
