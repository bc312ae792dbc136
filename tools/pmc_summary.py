#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite database."""
import sqlite3
import sys


def main(path, filt=None):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    rows = cur.execute("select %s, counter_name, avg(value), count(*) from counters_collection group by %s, counter_name" % (kn, kn)).fetchall()
    by = {}
    for k, c, v, n in rows:
        if filt and filt not in k:
            continue
        by.setdefault(k, {})[c] = (v, n)
    for k, d in by.items():
        print("### `%s`" % k[:100])
        for c in sorted(d):
            print("- %s = %.6g  (avg over %d dispatches)" % (c, d[c][0], d[c][1]))
        print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
