"""Turns the two pmc passes of tools/pmc_calib.py into calibration factors (known bytes / counter bytes)."""
import re, sqlite3, sys
KNOWN = {  # kernel substring -> (read MB, write MB)
    "relu_mask": (616.6, 308.3), "add2": (616.6, 308.3), "direct_copy": (308.3, 308.3), "upsample_softmax": (5.7, 359.5),
}
def per_kernel(dbpath, counter):
    db = sqlite3.connect(dbpath)
    out = {}
    for name, did, v in db.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name=? group by name, dispatch_id", (counter,)):
        a = out.setdefault(name, [0.0, 0]); a[0] += v; a[1] += 1
    return out
f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
print("| kernel | launches | known read MB | FETCH_SIZE MB (KiB x 1024) | known / counter | known write MB | WRITE_SIZE MB | known / counter |")
print("|---|---|---|---|---|---|---|---|")
for sub, (kr, kw) in KNOWN.items():
    for name in f:
        if sub in name:
            fr = f[name][0] / f[name][1] * 1024 / 1e6
            wr = w.get(name, [0, 1])[0] / max(w.get(name, [0, 1])[1], 1) * 1024 / 1e6
            print("| `{}` | {} | {:.1f} | {:.1f} | {:.2f} | {:.1f} | {:.1f} | {:.2f} |".format(re.sub(r"\(.*", "", name)[:60], f[name][1], kr, fr, kr / max(fr, 1e-9), kw, wr, kw / max(wr, 1e-9)))
