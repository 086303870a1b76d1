#!/bin/bash
# same-box A/B of environment switches: profiles/ab_env.sh "A=1" "B=2 C=3" ...  (use X=base for the default)
cd /root/repo
for round in 1 2; do
for cfg in "$@"; do
	env $cfg python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['pdus_in_timed_region'], d['pdus_rank0_matching_sent_payload'])"
done; done
