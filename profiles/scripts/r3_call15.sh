#!/bin/bash
# does the number of hardware queues explain the upload-rate lottery?  (streams A..E share 4 HW queues by default)
cd /root/repo
for q in 4 8; do
	echo "GPU_MAX_HW_QUEUES=$q"
	GPU_MAX_HW_QUEUES=$q python profiles/pcie_probe.py 2>/dev/null | tail -1
	GPU_MAX_HW_QUEUES=$q python profiles/pcie_probe.py --torch-first 2>/dev/null | tail -1
	GPU_MAX_HW_QUEUES=$q python bench.py --workload cfg2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bench cfg2: value %.0f host_ram %.0f host_path %.0f' % (r['value'], r['value_host_ram'], r['host_path']['value']))"
done
