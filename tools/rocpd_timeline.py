#!/usr/bin/env python
"""Timeline of the last few steps of a rocprofv3 kernel trace (rocpd sqlite): start / end of every kernel relative to the
first one shown, and the gap to the previous kernel's end.   usage: python tools/rocpd_timeline.py x_results.db [n_kernels]"""
import sqlite3
import sys


def main(path, n=40):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute("select %s, start, end, %s from kernels order by start" % (namecol, qcol)).fetchall()
    try:      # memory copies, if traced (named by direction)
        mc = [r[1] for r in cur.execute("pragma table_info(memory_copies)")]
        if mc:
            rows += [("copy:" + str(r[0]), r[1], r[2], -1) for r in cur.execute("select name, start, end from memory_copies").fetchall()]
            rows.sort(key=lambda r: r[1])
    except sqlite3.Error:
        pass
    rows = rows[-n:]
    t0 = rows[0][1]
    prev_end = None
    print("| kernel | queue | start us | end us | duration us | gap to the latest earlier end us |")
    print("|---|---|---|---|---|---|")
    for nm, s, e, q in rows:
        gap = "" if prev_end is None else "%.1f" % ((s - prev_end) / 1e3)
        print("| `%s` | %s | %.1f | %.1f | %.1f | %s |" % (nm[:40], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap))
        prev_end = e if prev_end is None else max(prev_end, e)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
