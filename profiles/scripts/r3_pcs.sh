#!/bin/bash
# PC sampling of the cfg2 workload: where does the demodulator's carrier wave spend its cycles?
cd /root/repo
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pcs
mkdir -p $OUT
for method in stochastic host_trap; do
	unit=cycles; ival=16384
	[ $method = host_trap ] && unit=time && ival=100
	timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $ival --kernel-trace \
		--output-format csv -d $OUT/$method -- python bench.py --workload cfg2 --steps 128 --no-cpu-baseline --no-extra-legs > $OUT/$method.log 2>&1
	echo "$method rc=$?"
	tail -5 $OUT/$method.log
	find $OUT/$method -type f | head -20
	f=$(find $OUT/$method -name "*pc_sampling*csv" | head -1)
	if [ -n "$f" ]; then wc -l $f; head -3 $f; break; fi
done
# keep what fits: compress the sample tables
find $OUT -name "*.csv" -size +1M -exec gzip -f {} \;
du -sh $OUT
