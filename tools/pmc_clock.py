"""Per-kernel clock (GRBM_GUI_ACTIVE / duration), matrix-pipe busy fraction and VALU:MFMA ratio from a
rocprofv3 --pmc rocpd database."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, counter_name, dispatch_id, sum(counter_value), count(*), max(duration) from pmc_events "
                  "where name like '%dasac%' group by name, counter_name, dispatch_id").fetchall()
agg = {}
for name, cn, did, v, n, dur in rows:
    k = re.sub(r"\(.*", "", name).replace("void dasac::", "")
    a = agg.setdefault(k, {}).setdefault(cn, [0.0, 0, 0.0, 0])
    a[0] += v; a[1] += 1; a[2] += dur; a[3] = n
print("{:44s} {:>6s} {:>9s} {:>9s} {:>10s} {:>9s}".format("kernel", "calls", "ms", "clk GHz", "mfma busy", "valu/mfma"))
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [0, 0, 0, 0])[2]):
    g = d.get("GRBM_GUI_ACTIVE")
    if not g:
        continue
    m, iv, im = d.get("SQ_VALU_MFMA_BUSY_CYCLES"), d.get("SQ_INSTS_VALU"), d.get("SQ_INSTS_MFMA")
    inst = g[3]                      # rows per dispatch (instances the counter is sampled on)
    clk = (g[0] / inst) / g[2]       # cycles per ns
    busy = (m[0] / (m[3] * 1.0)) / (g[0] / inst) / (1024.0 / m[3]) if m else float("nan")
    ratio = (iv[0] - im[0]) / im[0] if iv and im and im[0] else float("nan")
    print("{:44s} {:6d} {:9.2f} {:9.2f} {:10.3f} {:9.2f}".format(k[:44], g[1], g[2] / 1e6, clk, busy, ratio))
