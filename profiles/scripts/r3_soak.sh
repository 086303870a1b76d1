#!/bin/bash
# round-3 sustained runs: the bench loop over thousands of blocks (batching on cfg2), and the C host program for a short and a six times longer run
cd /root/repo
OUT=gpurun_out/r3soak; mkdir -p $OUT
for spec in "cfg3 6000" "cfg4 6000" "cfg2 40000"; do
set -- $spec
timeout 600 python bench.py --workload $1 --steps $2 --no-cpu-baseline --no-extra-legs 2>/dev/null > $OUT/soak_$1.json
python -c "import sys,json; r=json.load(open('$OUT/soak_$1.json')); print('$1', r['steps'], round(r['value'],1), r['pdus_in_timed_region'], r['pdus_matching_sent_payload'], r['pdus_lpdu_walk_matching_sent'], r['pdus_rank0_fcs_good_on_device'], round(r['roofline']['frac'],4), r['demod_blocks_per_launch'])"
done
bash profiles/scripts/host_soak.sh
