#!/bin/bash
# HBM traffic of the fold kernel: separate counter passes (FETCH_SIZE, WRITE_SIZE, L2 hit/miss), kernel trace only.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_*
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
	d=/tmp/pmc_$(echo $c | tr ' ' '_')
	rocprofv3 --pmc $c --kernel-trace -d $d -- python /root/repo/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $d.log 2>&1
done
python /root/repo/profiles/pmc_summary.py $(find /tmp/pmc_* -name "*.db" | sort)
