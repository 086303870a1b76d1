#!/bin/bash
# arithmetic PSK slicer against the lane-max one: cfg2 A/B, the slicer entry-point test, the parity suites, the SNR sweep
OUT=/root/repo/gpurun_out/r3j; mkdir -p $OUT
cd /root/repo
for i in 1 2 3; do
	HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lanemax.so timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs 2>/dev/null > $OUT/cfg2_lanemax_$i.json
	timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs 2>/dev/null > $OUT/cfg2_arith_$i.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r3j/cfg2_*.json")):
    r = json.load(open(f)); print(f.split("/")[-1], "value %.0f steady %.4f demod/blk %.4f pdus %d/%d" % (r["value"], r["steady_state_ms_per_step"], r["demod_kernel_ms_per_block"], r["pdus_matching_sent_payload"], r["pdus_in_timed_region"]))
PY
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
python - <<'PY'
import json
d = json.load(open("/root/repo/gpurun_out/low_snr_sweep.json"))
for r in d["rows"]:
    print(r["snr_db"], "gpu %d ora %d gpu_only %d identical %s recovered %d/%d same %s" % (r["gpu_pdus"], r["oracle_pdus"], r["gpu_only"], r["identical"], r["gpu_recovered"], r["oracle_recovered"], r["recovered_sets_identical"]))
PY
