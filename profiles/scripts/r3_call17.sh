#!/bin/bash
# dynamic instruction mix of the demodulator on cfg2 (what the kernel that no roofline describes actually executes): separate --pmc passes
OUT=/root/repo/gpurun_out/r3i; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" \
         "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SENDMSG SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM"; do
	i=$((i+1)); rm -rf /tmp/dm_$i
	timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/dm_$i -- python /root/repo/bench.py --workload cfg2 --steps 16 --warmup 4 --no-cpu-baseline --no-extra-legs > /tmp/dm_$i.log 2>&1
done
python /root/repo/profiles/pmc_summary.py $(find /tmp/dm_* -name "*.db" | sort) | grep "demod_kernel\|^| kernel\|^|---" > $OUT/cfg2_demod_inst_mix.md
cat $OUT/cfg2_demod_inst_mix.md
