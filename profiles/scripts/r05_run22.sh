#!/bin/bash
# round 5, run 22: v_mfma_f32_16x16x4_f32 -- lane layout and the arithmetic of one instruction (profiles/micro/mfma_k4.hip)
mkdir -p gpurun_out/r5u
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value profiles/micro/mfma_k4.hip -o /tmp/mfma_k4 && /tmp/mfma_k4 > gpurun_out/r5u/mfma_k4.txt 2>&1
cat gpurun_out/r5u/mfma_k4.txt
