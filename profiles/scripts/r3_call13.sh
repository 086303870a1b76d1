#!/bin/bash
cd /root/repo
t0=$(date +%s.%N); python bench.py > /tmp/b.json 2>/tmp/b.err; t1=$(date +%s.%N); echo "default bench wall $(echo "$t1 - $t0" | bc) s"
python -c "
import json; r=json.load(open('/tmp/b.json')); print('value %.0f steps %d'%(r['value'], r['steps'])); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in r['setup_s'].items()})"
t0=$(date +%s.%N); python bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/b2.json 2>/tmp/b2.err; t1=$(date +%s.%N); echo "bench --steps 20 --warmup 5 wall $(echo "$t1 - $t0" | bc) s"
python -c "
import json; r=json.load(open('/tmp/b2.json')); print('value %.0f steps %d ms %.4f'%(r['value'], r['steps'], r['ms_per_step']))"
