"""Per-kernel averages of every counter in a rocprofv3 --pmc results database (summed over XCDs / instances per dispatch,
averaged over dispatches) next to the average kernel duration.  Usage: python tools/pmc_table.py p_results.db"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
pmc = [t for t in tables if t.startswith("pmc_events") or t == "pmc_events"][0]
rows = {}
for name, counter, did, v in db.execute("select name, counter_name, dispatch_id, sum(counter_value) from %s group by name, counter_name, dispatch_id" % pmc):
    k = re.sub(r"\(.*", "", name).replace("void ", "").replace("dasac::", "")
    a = rows.setdefault(k, {}).setdefault(counter, [0.0, 0])
    a[0] += v
    a[1] += 1
dur = {}
kd = [t for t in tables if t.startswith("kernels") or t == "kernels"]
if kd:
    try:
        for name, d in db.execute("select name, avg(end - start) from %s group by name" % kd[0]):
            dur[re.sub(r"\(.*", "", name).replace("void ", "").replace("dasac::", "")] = d
    except Exception:
        pass
counters = sorted({c for r in rows.values() for c in r})
print("| kernel | launches | avg us | " + " | ".join(counters) + " |")
print("|---|---|---|" + "---|" * len(counters))
for k in sorted(rows, key=lambda k: -sum(v[0] for v in rows[k].values())):
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    n = max(v[1] for v in rows[k].values())
    print("| `{}` | {} | {} | ".format(k[:48], n, "%.1f" % (dur[k] / 1e3) if k in dur else "-") + " | ".join("%.4g" % (rows[k][c][0] / rows[k][c][1]) if c in rows[k] else "-" for c in counters) + " |")
