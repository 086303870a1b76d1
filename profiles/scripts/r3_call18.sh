#!/bin/bash
# several independent cfg2 receivers (one hfdl_replay process each) sharing ONE MI355X: aggregate rate, cs16 and cf32 files
cd /root/repo
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, bench, os
import dumphfdl_amd as hf
w = bench.WORKLOADS["cfg2"]
g = hf.plan_geometry(1024, 250 / w["fs"])
x, _ = bench.make_input(w, g.input_size, 0, 1)
d = "/dev/shm" if os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize > 1e9 else "/tmp"
x.view(np.float32).tofile(d + "/cfg2.cf32")
np.clip(np.round(x.view(np.float32) * 20000), -32768, 32767).astype(np.int16).tofile(d + "/cfg2.cs16")
open("/tmp/cfg2.dir", "w").write(d)
open("/tmp/cfg2.freqs", "w").write(" ".join("%.3f" % (f / 1e3) for f in bench.channel_plan(w)))
PY
D=$(cat /tmp/cfg2.dir)
for fmt in cs16 cf32; do
	F=$(echo $fmt | tr a-z A-Z)
	for k in 1 2 4 8; do
		rm -f /tmp/mr_*.json
		for i in $(seq 1 $k); do
			HFDL_FILE_READERS=4 /root/repo/dumphfdl_amd/hfdl_replay --bench --loop 300 --iq-file $D/cfg2.$fmt --sample-rate 8000000 --sample-format $F --centerfreq 10000.000 $(cat /tmp/cfg2.freqs) > /tmp/mr_$i.json 2>/dev/null &
		done
		wait
		python - <<PY
import json, glob
rs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("/tmp/mr_*.json"))]
print("$fmt x $k receivers: aggregate %.0f Msamples/s (%s), pdus %s" % (sum(r["value"] for r in rs), " ".join("%.0f" % r["value"] for r in rs), [r["pdus"] for r in rs]))
PY
	done
done
