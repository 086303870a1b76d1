# a minute of the C host path on each geometry: throughput, PDUs and the process's peak memory (leak check: two lengths)
cd /root/repo
python - <<'PY'
import json, os, subprocess, sys, resource
sys.path.insert(0, "/root/repo")
import numpy as np, bench
import dumphfdl_amd as hf
exe = "/root/repo/dumphfdl_amd/hfdl_replay"
for name, fmt in (("cfg3", "CS16"), ("cfg2", "CF32")):
    w = bench.WORKLOADS[name]
    g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
    x, _ = bench.make_input(w, g.input_size, 0, 1)
    raw = np.clip(np.round(x.view(np.float32) * 20000), -32768, 32767).astype(np.int16) if fmt == "CS16" else x.view(np.float32)
    path = "/dev/shm/soak_%s.%s" % (name, fmt.lower())
    raw.tofile(path)
    base = int(np.ceil(2.5e9 / len(x)))
    for mult in (4, 24):
        loops = base * mult
        cmd = [exe, "--bench", "--loop", str(loops), "--iq-file", path, "--sample-rate", str(w["fs"]), "--sample-format", fmt,
               "--centerfreq", "%.3f" % (w["centerfreq"] / 1e3)] + ["%.3f" % (f / 1e3) for f in bench.channel_plan(w)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=400)
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss          # kB, the largest child so far: the long run follows the short one
        print(name, fmt, "loops", loops, "blocks", r["blocks"], "seconds", round(r["seconds"], 2), "Msamples/s", round(r["value"], 1), "pdus", r["pdus"],
              "pdus per loop", round(r["pdus"] / loops, 2), "max RSS MB", rss // 1024)
    os.remove(path)
PY
