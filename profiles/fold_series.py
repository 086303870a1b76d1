#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (rocpd sqlite): the duration of every fold / demodulator launch in order, to see
launch-to-launch spread inside one run."""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
fold = [(e - s) / 1e3 for n, s, e in rows if "fold_kernel<" in n]
demod = [(e - s) / 1e3 for n, s, e in rows if "demod_kernel" in n]
k5 = [(e - s) / 1e3 for n, s, e in rows if "burst_decode" in n]
print("fold  us:", " ".join("%.0f" % v for v in fold))
print("demod us:", " ".join("%.0f" % v for v in demod))
print("k5    us:", " ".join("%.0f" % v for v in k5))
