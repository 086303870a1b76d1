#!/bin/bash
# HEAD verification: what the driver runs at round end
cd /root/repo
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read())
print('value %.0f ms %.4f frac %.4f traffic_commit %s host_ram %.0f cfg2 %.0f parity %s cpu %.1f keys %d' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic_source']['measured_at_commit'], r['value_host_ram'], r['cfg2']['value'], r['parity']['pdu_multisets_identical'], r['cpu_baseline']['value'], len(r)))"
