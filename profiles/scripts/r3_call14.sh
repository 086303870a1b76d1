#!/bin/bash
# the gloo fall-back: two ranks asked for RCCL on ONE GPU (RCCL refuses two ranks on a device)
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 2 --workload cfg2 --steps 26 --warmup 0 --backend nccl --no-cpu-baseline --no-extra-legs > /tmp/fb.out 2> /tmp/fb.err
echo "rc=$?"
grep -c "" /tmp/fb.out
python -c "
import json
r=json.loads(open('/tmp/fb.out').read().strip().splitlines()[-1]); print(r['distributed']); print(r['n_gpus'], r['value'], r['pdus_in_timed_region'], r['pdus_matching_sent_payload'], [(p['rank'], p['ms_per_step']) for p in r['per_rank']])"
tail -5 /tmp/fb.err | cut -c1-300
