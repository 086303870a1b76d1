#!/usr/bin/env python3
"""Where a demodulator-bound run loses time: the idle gaps between consecutive demod_kernel dispatches over a whole rocprofv3 trace
(rocpd sqlite, --kernel-trace [--memory-copy-trace]) -- histogram, the largest ones, and every event around the three largest."""
import sqlite3
import sys


def load(db):
    cur = sqlite3.connect(db).cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    for name, st, en, qq in cur.execute("select name, start, end, %s from kernels" % q):
        ev.append((st, en, name.split("(")[0].replace("void ", "").split("<")[0].replace("hfdl::", ""), "q%s" % qq))
    for t in tables:
        if "memory_cop" in t and "rocpd_" not in t:
            cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
            name = "name" if "name" in cols else "'copy'"
            for nm, st, en in cur.execute("select %s, start, end from %s" % (name, t)):
                ev.append((st, en, str(nm).replace("MEMORY_COPY_", ""), "dma"))
            break
    ev.sort()
    return ev


def main(db):
    ev = load(db)
    dm = [e for e in ev if e[2] == "demod_kernel"]
    if len(dm) < 3:
        print("no demodulator launches")
        return
    gaps = [(dm[i][0] - dm[i - 1][1], i) for i in range(1, len(dm))]
    busy = sum(e[1] - e[0] for e in dm)
    span = dm[-1][1] - dm[0][0]
    print("demod launches %d, span %.1f ms, busy %.1f ms (%.1f %%), mean launch %.1f us" % (len(dm), span / 1e6, busy / 1e6, 100.0 * busy / span, busy / len(dm) / 1e3))
    edges = [0, 15, 30, 60, 120, 250, 500, 1000, 2000, 5000, 1e9]
    for lo, hi in zip(edges, edges[1:]):
        sel = [g for g, _ in gaps if lo * 1e3 <= g < hi * 1e3]
        print("gap %5.0f .. %5.0f us: %4d launches, %.2f ms idle" % (lo, min(hi, 99999), len(sel), sum(sel) / 1e6))
    t0 = dm[0][0]
    top = sorted(gaps, reverse=True)[:8]
    print("largest gaps:", ", ".join("%.0f us @ %.1f ms" % (g / 1e3, (dm[i][0] - t0) / 1e6) for g, i in top))
    for g, i in top[:3]:
        a, b = dm[i - 1][1] - 300e3, dm[i][0] + 100e3
        print("\n-- around the %.0f us gap at %.1f ms --" % (g / 1e3, (dm[i][0] - t0) / 1e6))
        for st, en, nm, q in ev:
            if en >= a and st <= b:
                print("%-22s %-4s %9.1f %9.1f %8.1f" % (nm, q, (st - a) / 1e3, (en - a) / 1e3, (en - st) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
