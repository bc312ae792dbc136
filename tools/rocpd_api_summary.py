#!/usr/bin/env python
"""Per-call summary of the HIP API regions in a rocprofv3 (rocpd sqlite) trace taken with --hip-trace, plus the kernel table.
usage: python tools/rocpd_api_summary.py x_results.db [calls]   (calls: divide totals by this many host calls)"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
per = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cand = [n for n in names if n.lower() in ("regions", "region", "api")] or [n for n in names if "region" in n.lower()]
print("tables/views:", ", ".join(sorted(names))[:600])
for t in cand[:1]:
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
    print("using", t, cols)
    rows = cur.execute("select name, count(*), sum(end-start), max(end-start) from %s group by name order by 3 desc limit 30" % t).fetchall()
    print("| api | calls/host call | total us/host call | max us |")
    print("|---|---|---|---|")
    for n, c, tt, mx in rows:
        print("| %s | %.1f | %.1f | %.1f |" % (n, c / per, tt / 1e3 / per, mx / 1e3))
