"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share
a pass: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2).  Units: KiB.  gfx950 correction from
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports half of a wide coalesced read stream,
so the read side is doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import re, sqlite3, sys
def per_kernel(dbpath, counter):
    db = sqlite3.connect(dbpath)
    out = {}
    for name, did, v in db.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name=? group by name, dispatch_id", (counter,)):
        k = re.sub(r"\(.*", "", name).replace("void ", "")
        a = out.setdefault(k, [0.0, 0])
        a[0] += v; a[1] += 1
    return out
f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
print("| kernel | launches | FETCH_SIZE KiB/launch (raw) | read MB/launch (x2 gfx950) | WRITE_SIZE KiB/launch | write MB/launch | total MB/launch |")
print("|---|---|---|---|---|---|---|")
for k in sorted(f, key=lambda k: -f[k][0])[:12]:
    fr = f[k][0] / f[k][1]; wr = w.get(k, [0, 1])[0] / max(w.get(k, [0, 1])[1], 1)
    print("| `{}` | {} | {:.0f} | {:.1f} | {:.0f} | {:.1f} | {:.1f} |".format(k[:70], f[k][1], fr, 2 * fr * 1024 / 1e6, wr, wr * 1024 / 1e6, (2 * fr + wr) * 1024 / 1e6))
