#!/bin/bash
OUT=/root/repo/gpurun_out/r3f
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
rm -rf /tmp/tr
timeout 300 rocprofv3 --hip-trace --kernel-trace -d /tmp/tr -- /root/repo/dumphfdl_amd/hfdl_replay --bench --loop 20 --iq-file /tmp/cfg2.cf32 --sample-rate 8000000 --sample-format CF32 --centerfreq 10000.000 $(cat /tmp/cfg2.freqs) 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('traced', r['value'], r['blocks'], r['thread_s'], 'drains', r['pipeline_drains'])"
DB=$(find /tmp/tr -name "*.db" | head -1)
python /root/repo/profiles/host_stalls.py $DB 150 > $OUT/host_stalls.txt
head -100 $OUT/host_stalls.txt
