#!/bin/bash
# round 6, calls 8 - 9: the neighbour probe with 84 KiB of LDS per neighbour workgroup and 16 hardware queues (at 4 the neighbour stream shared a queue with one of the pipeline and every block waited for the neighbour to end)
# beside it and every block waited for the neighbour to end)
OUT=/root/repo/gpurun_out/r6p
mkdir -p $OUT
cd /root/repo
GPU_MAX_HW_QUEUES=16 timeout 900 python profiles/neighbour_probe.py cfg3 120 2> $OUT/neighbour.err | tee $OUT/neighbour_probe_cfg3.md | cut -c1-260
grep -v "amdgpu.ids\|UserWarning\|dev = torch" $OUT/neighbour.err | tail -n 5
