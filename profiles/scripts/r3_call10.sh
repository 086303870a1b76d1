#!/bin/bash
# file reader threads on this box: cfg2 cf32 through the C host program with 4 / 8 / 16 readers
cd /root/repo
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, bench
import dumphfdl_amd as hf
w = bench.WORKLOADS["cfg2"]
g = hf.plan_geometry(1024, 250 / w["fs"])
x, _ = bench.make_input(w, g.input_size, 0, 1)
import os
d = "/dev/shm" if os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize > 4e8 else "/tmp"
x.view(np.float32).tofile(d + "/cfg2.cf32")
open("/tmp/cfg2.dir", "w").write(d)
open("/tmp/cfg2.freqs", "w").write(" ".join("%.3f" % (f / 1e3) for f in bench.channel_plan(w)))
PY
D=$(cat /tmp/cfg2.dir); echo "file in $D; $(nproc) cpus"
for r in 4 8 16 8 16; do
	HFDL_FILE_READERS=$r /root/repo/dumphfdl_amd/hfdl_replay --bench --loop 150 --iq-file $D/cfg2.cf32 --sample-rate 8000000 --sample-format CF32 --centerfreq 10000.000 $(cat /tmp/cfg2.freqs) 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('readers $r', r['value'], r['thread_s'], 'drains', r['pipeline_drains'])"
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --workload cfg2 --steps 26 --warmup 0 --no-cpu-baseline --no-extra-legs 2>/dev/null | cut -c1-150
