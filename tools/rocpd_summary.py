#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel count / total / avg / min / max duration,
and PMC counter sums per kernel when present.  usage: rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    lines = []
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines.append("| kernel | calls | total ms | avg us | min us | max us | % |")
    lines.append("|---|---|---|---|---|---|---|")
    for name, n, s, a, mn, mx in rows:
        lines.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (name[:90], n, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    try:
        pmc = cur.execute(
            "select k.name, p.counter_name, count(*), sum(p.counter_value), avg(p.counter_value) from pmc_events p "
            "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
        if pmc:
            lines.append("")
            lines.append("| kernel | counter | dispatches | sum | avg per dispatch |")
            lines.append("|---|---|---|---|---|")
            for name, c, n, s, a in pmc:
                lines.append("| %s | %s | %d | %.6g | %.6g |" % (name[:60], c, n, s, a))
    except sqlite3.Error as e:
        lines.append("(no pmc data: %s)" % e)
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
