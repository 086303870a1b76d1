cd /root/repo
for wl in cfg2 cfg4; do timeout 200 python profiles/phase_probe.py $wl 2>&1 | grep alone | tail -2; done
