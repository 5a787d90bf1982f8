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
# Per training step (round 6): wall time on the device timeline from the step's first kernel (label_pad_mask of the SOURCE forward: the
# first launch of an iteration) to the next step's, the sum of kernel durations inside it, and the difference -- the "gap term":
# launch boundaries, idle time behind host work, anything that is not a kernel.  The last traced step has no successor and is skipped.
disp = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, (n, _, _) in enumerate(disp) if "label_pad_mask" in n]
# an iteration issues label_pad_mask twice (source labels, target labels): step boundaries are every SECOND mark from the first one
# that is followed by sgd_chunks before the next boundary
sgd = [i for i, (n, _, _) in enumerate(disp) if "sgd_chunks" in n]
bounds = []
for j, i in enumerate(sgd):                       # the step that ends with this optimiser launch starts at the last mark pair before it
    prev = sgd[j - 1] if j else -1
    ms_ = [m for m in marks if prev < m < i]
    if ms_:
        bounds.append(ms_[0])
if len(bounds) >= 2:
    lines.append("")
    lines.append("| training step | wall ms (first kernel to the next step's first kernel) | sum of kernel durations ms | gap ms | dispatches |")
    lines.append("|---|---|---|---|---|")
    for k, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
        wall = (disp[b][1] - disp[a][1]) / 1e6
        busy = sum(e - s_ for _, s_, e in disp[a:b]) / 1e6
        lines.append("| {} | {:.2f} | {:.2f} | {:.2f} | {} |".format(k, wall, busy, wall - busy, b - a))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "a").write(out + "\n")
