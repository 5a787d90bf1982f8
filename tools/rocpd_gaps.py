"""Idle time between kernels in a rocprofv3 (rocpd sqlite) kernel trace: over the last `frac` of the trace (the steady-state
steps), busy = union of kernel intervals, idle = span - busy, plus a histogram of the gaps and the kernels that follow the
longest ones.  Usage: python tools/rocpd_gaps.py <results.db> [frac=0.5]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = db.execute("select start, end, name from kernels order by start").fetchall()
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[0] >= cut]
span = rows[-1][1] - rows[0][0]
busy, cur_end, gaps = 0, rows[0][0], []
for s, e, name in rows:
    if s > cur_end:
        gaps.append((s - cur_end, name))
        busy += e - s
        cur_end = e
    else:
        busy += max(0, e - cur_end)
        cur_end = max(cur_end, e)
print("window {:.1f} ms, {} kernels: busy {:.1f} ms, idle {:.1f} ms ({:.2f} %), {} gaps, mean gap {:.2f} us".format(
    span / 1e6, len(rows), busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, len(gaps), sum(g for g, _ in gaps) / max(len(gaps), 1) / 1e3))
hist = collections.Counter()
for g, _ in gaps:
    hist["<2us" if g < 2e3 else "2-5us" if g < 5e3 else "5-20us" if g < 2e4 else "20-100us" if g < 1e5 else ">100us"] += g
for k in ("<2us", "2-5us", "5-20us", "20-100us", ">100us"):
    print("  gaps {:9s} total {:8.2f} ms".format(k, hist[k] / 1e6))
by = collections.Counter()
for g, n in gaps:
    by[re.sub(r"\(.*", "", n).replace("void ", "")[:60]] += g
print("idle time in front of (top 12):")
for n, g in by.most_common(12):
    print("  {:8.2f} ms  {}".format(g / 1e6, n))
# the longest individual gaps with their neighbours
rows2 = rows
long_gaps = []
cur_end, prev_name = rows2[0][0], None
for s, e, name in rows2:
    if s > cur_end and prev_name is not None:
        long_gaps.append((s - cur_end, (cur_end - rows2[0][0]) / 1e6, prev_name, name))
    if e >= cur_end:
        cur_end, prev_name = e, name
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("at::native::", "")[:48]
print("longest gaps (us, at ms, after kernel -> before kernel):")
for g, at, a, b in sorted(long_gaps, reverse=True)[:24]:
    print("  {:9.1f} us @ {:8.1f} ms  {}  ->  {}".format(g / 1e3, at, short(a), short(b)))
