#!/bin/bash
# round 3, fourth GPU call: timeline (kernels + copies) of the C host program on cfg2, the low-SNR test with its measured bounds,
# the libm-trig experimental build on the bins below +2 dB.
OUT=/root/repo/gpurun_out/r3d
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_low_snr.py -m gpu -q > $OUT/pytest_low.log 2>&1; echo "rc=$?" >> $OUT/pytest_low.log
tail -4 $OUT/pytest_low.log
cp gpurun_out/low_snr_sweep.json $OUT/low_snr_default.json 2>/dev/null
HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_libm.so timeout 600 python profiles/low_snr_parity.py --bins=-8:2:2 > $OUT/low_snr_libm.json 2> $OUT/low_snr_libm.err
python - <<'PY'
import json
for f in ("low_snr_default", "low_snr_libm"):
    try:
        d = json.load(open("/root/repo/gpurun_out/r3d/%s.json" % f))
        for r in d["rows"]:
            print(f, r["snr_db"], "gpu %d ora %d common %d gpu_only %d ora_only %d identical %s recovered %d/%d same %s changed %d moved %d" % (
                r["gpu_pdus"], r["oracle_pdus"], r["common"], r["gpu_only"], r["oracle_only"], r["identical"], r["gpu_recovered"], r["oracle_recovered"],
                r["recovered_sets_identical"], r["same_place_other_octets"], r["same_octets_other_place"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
# the C host program under a kernel + copy trace
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, bench
import dumphfdl_amd as hf
w = bench.WORKLOADS["cfg2"]
g = hf.plan_geometry(1024, 250 / w["fs"])
x, _ = bench.make_input(w, g.input_size, 0, 1)
x.view(np.float32).tofile("/tmp/cfg2.cf32")
np.clip(np.round(x.view(np.float32) * 20000), -32768, 32767).astype(np.int16).tofile("/tmp/cfg2.cs16")
open("/tmp/cfg2.freqs", "w").write(" ".join("%.3f" % (f / 1e3) for f in bench.channel_plan(w)))
PY
cd /tmp && export TMPDIR=/tmp
for fmt in cf32 cs16; do
	F=$(echo $fmt | tr a-z A-Z)
	/root/repo/dumphfdl_amd/hfdl_replay --bench --loop 40 --iq-file /tmp/cfg2.$fmt --sample-rate 8000000 --sample-format $F --centerfreq 10000.000 $(cat /tmp/cfg2.freqs) > $OUT/replay_$fmt.json 2>&1
	rm -rf /tmp/tr_$fmt
	timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tr_$fmt -- /root/repo/dumphfdl_amd/hfdl_replay --bench --loop 20 --iq-file /tmp/cfg2.$fmt --sample-rate 8000000 --sample-format $F --centerfreq 10000.000 $(cat /tmp/cfg2.freqs) > $OUT/replay_${fmt}_traced.json 2>&1
	DB=$(find /tmp/tr_$fmt -name "*.db" | head -1)
	python /root/repo/profiles/timeline_all.py $DB 90 30 > $OUT/replay_${fmt}_timeline.md
done
tail -1 $OUT/replay_cf32.json | cut -c1-400
tail -1 $OUT/replay_cs16.json | cut -c1-400
head -70 $OUT/replay_cf32_timeline.md
