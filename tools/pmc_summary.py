"""Aggregates rocprofv3 --pmc results (rocpd sqlite) per kernel: mean counter value per dispatch."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else "conv"
cols = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
q = "select * from pmc_events limit 1"
rows = db.execute("select name from sqlite_master where name like 'pmc_events'").fetchall()
try:
    data = db.execute("select name, counter_name, avg(counter_value), count(*), avg(duration) from pmc_events group by name, counter_name").fetchall()
except Exception as e:
    print("schema:", cols, e)
    raise
for name, cn, v, n, dur in data:
    if filt in name:
        print("{:50s} {:28s} {:16.1f} (n={}, avg dur {:.1f} us)".format(re.sub(r"\(.*", "", name)[:50], cn, v, n, dur / 1e3))
