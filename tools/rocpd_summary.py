#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd sqlite database --
what `rocprofv3 --kernel-trace --stats` prints, as a CSV that can be committed under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    # one row per (kernel, grid): the same GEMM kernel serves launches of very different sizes
    grid = "grid_x * grid_y * grid_z" if "grid_x" in cols else "0"
    rows = c.execute("select %s, %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                     "group by %s, %s order by 4 desc" % (name_col, grid, name_col, grid)).fetchall()
    total = sum(r[3] for r in rows) or 1
    lines = ["Name,GridThreads,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for n, g, calls, tot, avg, mn, mx in rows:
        lines.append('"%s",%d,%d,%d,%.1f,%.2f,%d,%d' % (n, g, calls, tot, avg, 100.0 * tot / total, mn, mx))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
