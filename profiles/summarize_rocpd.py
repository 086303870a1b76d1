#!/usr/bin/env python3
"""Turn a rocprofv3 (--kernel-trace --stats, rocpd sqlite output) database into the per-kernel summary table kept in profiles/."""
import sqlite3
import sys


fft_points = None


def main(db_path, title):
    global fft_points
    for key, n in (("cfg3", 1 << 23), ("cfg4", 1 << 23), ("cfg2", 1 << 20)):
        if title.startswith(key):
            fft_points = n
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                       "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# %s\n" % title)
    # forward-FFT passes: every pass reads and writes the whole N-point buffer once (8 N bytes each way); N from the grid
    # (pass 1 / 2: N / 16 columns-of-16 tiles x 512 threads; pass 3 likewise) -- GB/s against the 8 TB/s HBM peak per pass
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | grid | wg | HBM GB/s (min-time launch) | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    nfft = None
    for r in rows:
        if "fft_pass3" in r[0]:
            nfft = (r[9] // r[10]) * 16 * 128 if False else None
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")
        gbs = ""
        frac = ""
        if ("fft_pass" in name or "fft_rpass" in name) and fft_points:
            b = 16.0 * fft_points
            gbs = "%.0f" % (b / (r[4] * 1e-6) / 1e9)
            frac = "%.2f" % (b / (r[4] * 1e-6) / 1e9 / 8000.0)
        print("| %s | %d | %.2f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s | %s | %s |" % (name, r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], gbs, frac))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
