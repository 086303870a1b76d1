#!/bin/bash
# round 3, third GPU call: host -> device copies on 1 / 2 / 4 streams (cfg2 and cfg3, Python host-RAM leg and the C host program),
# the low-SNR sweep (test + the libm-trig experimental build), and the two new tests.
OUT=/root/repo/gpurun_out/r3c
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_low_snr.py tests/test_gpu_parity.py -m gpu -q -k "low_snr or marginal or prefetch or batching" > $OUT/pytest_sel.log 2>&1; echo "rc=$?" >> $OUT/pytest_sel.log
tail -15 $OUT/pytest_sel.log
cp gpurun_out/low_snr_sweep.json $OUT/low_snr_default.json 2>/dev/null
HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_libm.so timeout 600 python profiles/low_snr_parity.py --bins -8:2:2 > $OUT/low_snr_libm.json 2> $OUT/low_snr_libm.err
python - <<'PY'
import json
for f in ("low_snr_default", "low_snr_libm"):
    try:
        d = json.load(open("/root/repo/gpurun_out/r3c/%s.json" % f))
        for r in d["rows"]:
            print(f, r["snr_db"], "gpu %d ora %d common %d gpu_only %d ora_only %d identical %s recovered %d/%d changed %d moved %d" % (
                r["gpu_pdus"], r["oracle_pdus"], r["common"], r["gpu_only"], r["oracle_only"], r["identical"], r["gpu_recovered"], r["oracle_recovered"],
                r["same_place_other_octets"], r["same_octets_other_place"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
for wl in cfg2 cfg3; do
	for ways in 1 2 4; do
		HFDL_GPU_COPY_STREAMS=$ways timeout 300 python bench.py --workload $wl --host-input --steps 128 --no-cpu-baseline --no-extra-legs > $OUT/host_${wl}_ways$ways.json 2>> $OUT/bench.err
	done
done
python - > $OUT/host_path_ways.json 2>> $OUT/bench.err <<'PY'
import json, os, sys
sys.path.insert(0, "/root/repo")
import bench
import dumphfdl_amd as hf
out = {}
for name in ("cfg2", "cfg3"):
    w = bench.WORKLOADS[name]
    g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
    x, _ = bench.make_input(w, g.input_size, 0, 1)
    for ways in (1, 2, 4):
        os.environ["HFDL_GPU_COPY_STREAMS"] = str(ways)
        for fmt in ("CF32", "CS16"):
            r = bench.host_path_leg(w, x, bench.channel_plan(w), fmt)
            out["%s_%s_ways%d" % (name, fmt, ways)] = dict(value=r.get("value"), thread_s=r.get("thread_s"), error=r.get("error"))
print(json.dumps(out))
PY
python - <<'PY'
import json, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r3c/host_cfg*_ways*.json")):
    try:
        r = json.load(open(f)); print(f.split("/")[-1], "value %.0f ms/step %.4f" % (r["value"], r["ms_per_step"]))
    except Exception as e:
        print(f, "unreadable", e)
try:
    d = json.load(open("/root/repo/gpurun_out/r3c/host_path_ways.json"))
    for k, v in d.items(): print(k, v["value"], v["error"])
except Exception as e:
    print("host_path_ways unreadable", e)
PY
