#!/usr/bin/env python3
"""Which HIP API calls block the host thread, from a rocprofv3 --hip-trace --kernel-trace run (rocpd sqlite): per API name the count,
total and maximum duration, and a time-ordered list of the calls longer than `min_us` with the demodulator launches around them."""
import sqlite3
import sys


def main(db, min_us=150.0, limit=60):
    cur = sqlite3.connect(db).cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    api = None
    for t in ("regions", "hip_api", "api"):
        if t in tables:
            api = t
            break
    if api is None:
        cand = [t for t in tables if "region" in t and not t.startswith("rocpd_")]
        api = cand[0] if cand else None
    print("tables:", ", ".join(t for t in tables if not t.startswith("rocpd_")))
    if api is None:
        return
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % api)]
    print(api, cols)
    tid = "tid" if "tid" in cols else ("thread_id" if "thread_id" in cols else "0")
    rows = cur.execute("select name, start, end, %s from %s" % (tid, api)).fetchall()
    agg = {}
    for nm, st, en, t in rows:
        a = agg.setdefault(nm, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (en - st) / 1e3; a[2] = max(a[2], (en - st) / 1e3)
    print("| API | calls | total ms | max us |")
    for nm, (n, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("| %s | %d | %.2f | %.1f |" % (nm, n, tot / 1e3, mx))
    dm = cur.execute("select start, end from kernels where name like '%demod_kernel%' order by start").fetchall()
    t0 = dm[0][0] if dm else rows[0][1]
    ev = [((st - t0) / 1e3, "API  %-28s %8.1f us (tid %s)" % (nm, (en - st) / 1e3, t)) for nm, st, en, t in rows if (en - st) / 1e3 >= min_us]
    ev += [((st - t0) / 1e3, "GPU  demod_kernel                 %8.1f us" % ((en - st) / 1e3)) for st, en in dm]
    ev.sort()
    mid = len(ev) // 2
    for t, s in ev[mid:mid + limit]:
        print("%10.1f  %s" % (t, s))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 150.0)
