#!/usr/bin/env python
"""Timeline of one bench step from a rocprofv3 rocpd sqlite database: every kernel between two consecutive
prepare_all_kernel launches (start offset, duration, queue), to read launch gaps and cross-stream overlap."""
import sqlite3
import sys


def main(db, which=-3, window_us=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = "%s, start, end%s" % (name_col, (", " + qcol) if qcol else "")
    rows = c.execute("select %s from kernels order by start" % sel).fetchall()
    marks = []   # (a head-first model prepares in two launches, one per stream, at the head of a step: the first of a cluster marks the step)
    for i, r in enumerate(rows):
        if "prepare_all" in r[0] and (not marks or r[1] - rows[marks[-1]][1] > 40e3):
            marks.append(i)
    lo, hi = marks[int(which)], marks[int(which) + 1]
    t0 = rows[lo][1]
    if window_us:   # steps in flight overlap: every kernel that STARTS within the window, whichever step it belongs to
        sel_rows = [r for r in rows if t0 - 50e3 <= r[1] <= t0 + float(window_us) * 1e3]
    else:
        sel_rows = rows[lo:hi]
    for r in sel_rows:
        nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        print("%9.1f us  +%8.1f us  q=%s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", nm))
    print("step span: %.1f us" % ((rows[hi][1] - t0) / 1e3))


if __name__ == "__main__":
    main(*sys.argv[1:4])
