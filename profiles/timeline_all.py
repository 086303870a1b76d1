#!/usr/bin/env python3
"""Kernels AND memory copies of a rocprofv3 --kernel-trace --memory-copy-trace run (rocpd sqlite), merged by start time:
the last `rows` events before the end of the trace minus `skip_ms`, times in us relative to the first one printed."""
import sqlite3
import sys


def main(db, rows=80, skip_ms=20.0):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    if "kernels" in tables:
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
        for name, st, en, qq in cur.execute("select name, start, end, %s from kernels" % q):
            ev.append((st, en, name.split("(")[0].replace("void ", "").split("<")[0].replace("hfdl::", ""), "q%s" % qq, ""))
    for t in tables:
        if "memory_cop" in t and "rocpd_" not in t:
            cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
            size = "size" if "size" in cols else ("bytes" if "bytes" in cols else "0")
            name = "name" if "name" in cols else "'copy'"
            for nm, st, en, sz in cur.execute("select %s, start, end, %s from %s" % (name, size, t)):
                ev.append((st, en, str(nm).replace("MEMORY_COPY_", ""), "dma", "%.2f MB %.1f GB/s" % (sz / 1e6, sz / max(1, en - st)) if sz else ""))
            break
    ev.sort()
    if not ev:
        print("no events; tables:", tables)
        return
    t_end = ev[-1][0] - skip_ms * 1e6
    sel = [e for e in ev if e[0] <= t_end][-rows:]
    t0 = sel[0][0]
    print("| event | queue | start us | end us | dur us | note |")
    print("|---|---|---|---|---|---|")
    for st, en, nm, q, note in sel:
        print("| %s | %s | %.1f | %.1f | %.1f | %s |" % (nm, q, (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, note))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80, float(sys.argv[3]) if len(sys.argv) > 3 else 20.0)
