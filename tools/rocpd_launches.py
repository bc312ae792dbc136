#!/usr/bin/env python
"""Durations of one kernel's launches in a rocprofv3 kernel trace (rocpd sqlite), in launch order, averaged per phase of the traced command.
usage: python tools/rocpd_launches.py x_results.db KERNEL_SUBSTRING name1:n1 name2:n2 ...   (the last phase takes what is left)"""
import sqlite3
import sys


def main(path, sub, phases):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = [(s, e) for nm, s, e in cur.execute("select %s, start, end from kernels order by start" % namecol).fetchall() if sub in nm]
    print("| phase of the traced command | launches of `%s` | average us | min us | max us |" % sub)
    print("|---|---|---|---|---|")
    i = 0
    for k, ph in enumerate(phases):
        name, _, n = ph.partition(":")
        n = len(rows) - i if (k == len(phases) - 1 or not n) else int(n)
        d = [(e - s) / 1e3 for s, e in rows[i:i + n]]
        i += n
        if d:
            print("| %s | %d | %.1f | %.1f | %.1f |" % (name, len(d), sum(d) / len(d), min(d), max(d)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
