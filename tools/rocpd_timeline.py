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
    rows = cur.execute("select %s, start, end from kernels order by start" % namecol).fetchall()[-n:]
    t0 = rows[0][1]
    prev_end = None
    print("| kernel | start us | end us | duration us | gap to the latest earlier end us |")
    print("|---|---|---|---|---|")
    for nm, s, e in rows:
        gap = "" if prev_end is None else "%.1f" % ((s - prev_end) / 1e3)
        print("| `%s` | %.1f | %.1f | %.1f | %s |" % (nm[:40], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap))
        prev_end = e if prev_end is None else max(prev_end, e)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
