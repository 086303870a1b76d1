#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 --pmc runs (rocpd sqlite databases given on the command line)."""
import sqlite3
import sys

print("| kernel | counter | dispatches | avg | min | max |")
print("|---|---|---|---|---|---|")
for db in sys.argv[1:]:
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
         "where kernel_name like '%hfdl%' group by kernel_name, counter_name order by 2, 4 desc")
    for name, ctr, n, a, lo, hi in cur.execute(q):
        print("| %s | %s | %d | %.8g | %.8g | %.8g |" % (name.split("(")[0].replace("void ", ""), ctr, n, a, lo, hi))
