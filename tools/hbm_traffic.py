"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots,
FETCH_SIZE takes 3, WRITE_SIZE 2).  Units: KiB.  gfx950 correction from /opt/skills/guides/MI355X_MICROARCH.md (HBM section,
confirmed on known-byte kernels in profiles/r2_pmc_calibration.md): FETCH_SIZE reports half of a wide coalesced read stream, so
the read side is doubled; WRITE_SIZE is exact.

Usage: python tools/hbm_traffic.py <fetch.db> <write.db> [--json traffic.json]
Two tables: per kernel SYMBOL over the whole process, and per kernel CLASS (what bench.py's `kernels` / `roofline` report:
the template instantiations of one schedule together) over the TRAINING STEPS only -- dispatches from the first
`label_pad_mask` on (the first kernel of an iteration); bench.py's model set-up forwards are left out."""
import json
import re
import sqlite3
import sys


def load(dbpath, counter):
    db = sqlite3.connect(dbpath)
    rows = db.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name=? group by name, dispatch_id "
                      "order by dispatch_id", (counter,)).fetchall()
    return [(re.sub(r"\(.*", "", n).replace("void ", ""), d, v) for n, d, v in rows]


def klass(name):
    m = re.match(r"dasac::conv_gemm<([^>]*)>", name)
    if m:
        args = [a.strip() for a in m.group(1).split(",")]
        # template argument 6 = SK: 0 one block per tile, 1 persistent stream-K, 2 tile-per-block rounds + split-K tail (round 6;
        # rounds 1-5: a bool)
        return {"1": "conv_gemm<stream-K>", "true": "conv_gemm<stream-K>", "2": "conv_gemm<tile+tail>"}.get(args[5], "conv_gemm<tile-per-block>")
    if name.startswith("dasac::conv_wgrad<"):
        return "conv_wgrad"
    return None


def per(rows, key, start=None):
    out = {}
    for name, did, v in rows:
        if start is not None and did < start:
            continue
        k = key(name)
        if k is None:
            continue
        a = out.setdefault(k, [0.0, 0])
        a[0] += v
        a[1] += 1
    return out


frows, wrows = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
f, w = per(frows, lambda n: n), per(wrows, lambda n: n)
print("| kernel symbol (whole process) | launches | FETCH_SIZE KiB/launch (raw) | read MB/launch (x2 gfx950) | WRITE_SIZE KiB/launch | write MB/launch | total MB/launch |")
print("|---|---|---|---|---|---|---|")
for k in sorted(f, key=lambda k: -f[k][0])[:14]:
    fr = f[k][0] / f[k][1]
    wr = w.get(k, [0, 1])[0] / max(w.get(k, [0, 1])[1], 1)
    print("| `{}` | {} | {:.0f} | {:.1f} | {:.0f} | {:.1f} | {:.1f} |".format(k[:70], f[k][1], fr, 2 * fr * 1024 / 1e6, wr, wr * 1024 / 1e6, (2 * fr + wr) * 1024 / 1e6))


def first(rows, needle):
    for name, did, _ in rows:
        if needle in name:
            return did
    return None


fs, ws = first(frows, "label_pad_mask"), first(wrows, "label_pad_mask")
fc, wc = per(frows, klass, fs), per(wrows, klass, ws)
steps = max(1, sum(1 for n, d, _ in frows if "sgd_chunks" in n and (fs is None or d >= fs)))
print()
print("| kernel class (training steps only: {} step(s)) | launches/step | read MB/launch | write MB/launch | total MB/launch |".format(steps))
print("|---|---|---|---|---|")
out = {}
for k in sorted(fc, key=lambda k: -fc[k][0]):
    rd = 2 * fc[k][0] / fc[k][1] * 1024
    wr = wc.get(k, [0, 1])[0] / max(wc.get(k, [0, 1])[1], 1) * 1024
    print("| `{}` | {} | {:.1f} | {:.1f} | {:.1f} |".format(k, fc[k][1] // steps, rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
    out[k] = {"hbm_bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "write_bytes": round(wr), "launches_per_step": fc[k][1] // steps,
              "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the training steps of `python bench.py` (tools/profile_round.sh, "
                        "tools/hbm_traffic.py): all template instantiations of the class, dispatches from the first label_pad_mask on; read side x2 "
                        "(FETCH_SIZE reports half of a coalesced stream on gfx950, WRITE_SIZE is exact: profiles/r2_pmc_calibration.md)"}
if "--json" in sys.argv:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                      # csrc_sha1(): bench.py reports `traffic` only for the kernel sources it was measured on
    out["csrc_sha1"] = bench.csrc_sha1()
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
