"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
Usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for name, n, tot, avg, mn, mx in rows[:40]:
    short = re.sub(r"\(.*", "", name)
    short = short.replace("void ", "")[:110]
    lines.append("| `{}` | {} | {:.2f} | {:.1f} | {:.1f} | {:.1f} | {:.1f} |".format(short, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
lines.append("")
lines.append("total GPU kernel time: {:.2f} ms over {} dispatches".format(total / 1e6, sum(r[1] for r in rows)))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "a").write(out + "\n")
