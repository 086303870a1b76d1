#!/bin/bash
# round 5: whole GPU suite
OUT=/root/repo/gpurun_out/r5g
mkdir -p $OUT
cd /root/repo
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
python - <<PY
import json
try:
    for r in json.load(open("/root/repo/gpurun_out/low_snr_sweep.json"))["rows"]:
        print({k: r[k] for k in ("snr_db", "gpu_pdus", "oracle_pdus", "common", "gpu_only", "oracle_only", "same_place_other_octets", "identical", "gpu_recovered", "oracle_recovered")})
except Exception as e:
    print("no sweep", e)
PY
