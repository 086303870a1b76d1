#!/bin/bash
# round 5, run 17: traffic records (PMC passes) at the csrc of the pruned-fold commit; default mode only
C=$(python -c "import json; print(json.load(open('/root/repo/profiles/scripts/stamp.json'))['commit'])" 2>/dev/null || echo unknown)
OUT=/root/repo/gpurun_out/final3
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in cfg3 cfg2 cfg4; do
	bash /root/repo/profiles/pmc_passes.sh $wl $OUT $C > $OUT/pmc_$wl.log 2>&1
	cp $OUT/fold_traffic_$wl.json /root/repo/profiles/fold_traffic_$wl.json
	tail -2 $OUT/pmc_$wl.log
done
cp /root/repo/profiles/fold_traffic_*.json $OUT/
