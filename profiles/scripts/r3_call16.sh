#!/bin/bash
cd /root/repo
for p in 0 1; do
	echo "HFDL_GPU_COPY_PRIO=$p"
	HFDL_GPU_COPY_PRIO=$p python profiles/pcie_probe.py 2>/dev/null | tail -1
	HFDL_GPU_COPY_PRIO=$p python profiles/pcie_probe.py --torch-first 2>/dev/null | tail -1
	HFDL_GPU_COPY_PRIO=$p python bench.py --workload cfg2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bench cfg2: value %.0f host_ram %.0f host_path %.0f' % (r['value'], r['value_host_ram'], r['host_path']['value']))"
	HFDL_GPU_COPY_PRIO=$p python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bench cfg3: value %.0f host_ram %.0f host_path %.0f; cfg2 leg host_ram %.0f' % (r['value'], r['value_host_ram'], r['host_path']['value'], r['cfg2']['value_host_ram']))"
done
