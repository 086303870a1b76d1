#!/usr/bin/env python3
"""The last K kernels of a rocprofv3 kernel trace (rocpd sqlite), one line each: queue, start / end relative to the first of them,
duration -- the whole timed region of a short run (bench.py --steps 20: 16 + 4 blocks).  Consecutive launches of one kernel on one
queue are folded into one line (count, first start, last end, busy time).  K < 0: around the last two fold launches instead."""
import sqlite3
import sys


def main(db_path, k=100):
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
    folds = [i for i, r in enumerate(rows) if "fold_mfma" in r[0]]
    if len(folds) >= 2 and k < 0:           # k < 0: the run's last two fold launches with -k kernels before the first and after the second
        rows = rows[max(0, folds[-2] + k):folds[-1] - k]
    else:
        rows = rows[-k:]
    t0 = rows[0][1]
    out = []
    for name, st, en, q in rows:
        short = name.split("(")[0].replace("void ", "").split("<")[0].replace("hfdl::", "")
        if out and out[-1][0] == short and out[-1][1] == q:
            out[-1][2] += 1; out[-1][4] = en; out[-1][5] += en - st
        else:
            out.append([short, q, 1, st, en, en - st])
    print("| kernel | queue | launches | first start us | last end us | busy us |")
    print("|---|---|---|---|---|---|")
    for short, q, n, st, en, busy in out:
        print("| %s | %s | %d | %.0f | %.0f | %.0f |" % (short, q, n, (st - t0) / 1e3, (en - t0) / 1e3, busy / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100)
