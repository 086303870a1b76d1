cd /root/repo
for i in 1 2; do
timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg2', round(r['value'],1), round(r['steady_state_ms_per_step'],4), r['pdus_in_timed_region'], r['pdus_matching_sent_payload'])"
done
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg3', round(r['value'],1), round(r['steady_state_ms_per_step'],4), r['pdus_in_timed_region'], r['pdus_matching_sent_payload'], round(r['roofline']['frac'],4))"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
