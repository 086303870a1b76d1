#!/usr/bin/env python3
"""Turn a rocprofv3 (--kernel-trace --stats, rocpd sqlite output) database into the per-kernel summary table kept in profiles/."""
import sqlite3
import sys


def main(db_path, title):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                       "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# %s\n" % title)
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")
        print("| %s | %d | %.2f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s |" % (name, r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8], r[9], r[10]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
