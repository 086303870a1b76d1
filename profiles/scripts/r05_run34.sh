#!/bin/bash
# round 5, run 34: the whole GPU suite + smoke at the end-of-round tree, libraries rebuilt from scratch by __graft_entry__.build()
mkdir -p gpurun_out/r5af
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5af/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r5af/pytest_gpu.log
tail -4 gpurun_out/r5af/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
