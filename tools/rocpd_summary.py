#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.

usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_summary.md
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc"
        % (namecol, namecol)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for n, c, t, a, mn, mx in rows:
        print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (n[:90], c, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    extra = [c for c in cols if c.lower() in ("vgpr_count", "sgpr_count", "lds_block_size", "workgroup_size", "grid_size", "accum_vgpr_count", "scratch_size")]
    if extra:
        print()
        print("| kernel | " + " | ".join(extra) + " |")
        print("|---|" + "---|" * len(extra))
        for r in cur.execute("select %s, %s from kernels group by %s" % (namecol, ", ".join("max(%s)" % e for e in extra), namecol)):
            print("| `%s` | " % r[0][:90] + " | ".join(str(x) for x in r[1:]) + " |")


if __name__ == "__main__":
    main(sys.argv[1])
