#!/bin/bash
# where the C host program loses time against the resident-input rate on cfg2: demodulator gaps over a whole traced run
OUT=/root/repo/gpurun_out/r3e
mkdir -p $OUT
cd /root/repo
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, bench
import dumphfdl_amd as hf
w = bench.WORKLOADS["cfg2"]
g = hf.plan_geometry(1024, 250 / w["fs"])
x, _ = bench.make_input(w, g.input_size, 0, 1)
x.view(np.float32).tofile("/tmp/cfg2.cf32")
open("/tmp/cfg2.freqs", "w").write(" ".join("%.3f" % (f / 1e3) for f in bench.channel_plan(w)))
PY
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
	/root/repo/dumphfdl_amd/hfdl_replay --bench --loop 150 --iq-file /tmp/cfg2.cf32 --sample-rate 8000000 --sample-format CF32 --centerfreq 10000.000 $(cat /tmp/cfg2.freqs) 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['blocks'], r['thread_s'], 'drains', r['pipeline_drains'])"
done
rm -rf /tmp/tr
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tr -- /root/repo/dumphfdl_amd/hfdl_replay --bench --loop 60 --iq-file /tmp/cfg2.cf32 --sample-rate 8000000 --sample-format CF32 --centerfreq 10000.000 $(cat /tmp/cfg2.freqs) 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('traced', r['value'], r['blocks'], r['thread_s'], 'drains', r['pipeline_drains'])"
DB=$(find /tmp/tr -name "*.db" | head -1)
python /root/repo/profiles/demod_gaps.py $DB > $OUT/replay_gaps.txt
head -120 $OUT/replay_gaps.txt
