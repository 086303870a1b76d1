#!/usr/bin/env python3
"""Per-block timeline from a rocprofv3 kernel trace (rocpd sqlite): for the last few fold launches, the start/end of every
kernel between two consecutive fold starts, relative to the first, with the idle gap before each kernel on its queue."""
import sqlite3
import sys


def main(db_path, blocks=2):
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else ", 0")).fetchall()
    folds = [i for i, r in enumerate(rows) if ("fold_kernel" in r[0] or "fold_mfma" in r[0]) and "generic" not in r[0]]
    if len(folds) < blocks + 2:
        print("not enough fold launches")
        return
    i0, i1 = folds[-blocks - 1], folds[-1]
    t0 = rows[i0][1]
    last_end = {}
    print("| kernel | queue | start us | end us | dur us | gap before (same queue) us |")
    print("|---|---|---|---|---|---|")
    for name, st, en, q in rows[i0:i1 + 1]:
        short = name.split("(")[0].replace("void ", "").split("<")[0]
        gap = (st - last_end[q]) / 1e3 if q in last_end else float("nan")
        print("| %s | %s | %.1f | %.1f | %.1f | %.1f |" % (short, q, (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, gap))
        last_end[q] = en


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
