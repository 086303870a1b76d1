#!/bin/bash
cd /root/repo
for i in 1 2; do
timeout 300 python bench.py --workload cfg2 --no-cpu-baseline > /tmp/b2.json 2>/dev/null
python -c "
import json; r=json.load(open('/tmp/b2.json')); print('cfg2 main: value %.0f host_ram %.0f pcie %.1f GB/s host_path %.0f'%(r['value'], r['value_host_ram'], r['host_ram_input']['pcie_GBs'], r['host_path']['value']))"
done
timeout 600 python bench.py --no-cpu-baseline > /tmp/b3.json 2>/dev/null
python -c "
import json; r=json.load(open('/tmp/b3.json')); print('cfg3 main: value %.0f host_ram %.0f pcie %.1f GB/s host_path %.0f; cfg2 leg value %.0f host_ram %.0f pcie %.1f'%(r['value'], r['value_host_ram'], r['host_ram_input']['pcie_GBs'], r['host_path']['value'], r['cfg2']['value'], r['cfg2']['value_host_ram'], r['cfg2']['host_ram_input']['pcie_GBs']))"
