"""Instruction mix of the kernels in a device assembly listing: python tools/isa_count.py file.s [name substring ...].
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -Iinclude da-sac_amd/csrc/X.hip -o file.s)
Per kernel: total instructions, the hottest loop's body (between its label and its back edge), VALU / SALU / memory / LDS split, registers."""
import re
import sys
from collections import Counter


def kind(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "mem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2:]
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\n\s*\.amdhsa_kernel \1", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want and not any(w in name for w in want):
            continue
        lines = [l.strip() for l in body.split("\n")]
        ops, labels = [], {}
        for l in lines:
            if re.match(r"^\.LBB\d+_\d+:", l):
                labels[l.split(":")[0]] = len(ops)
            elif l and not l.startswith((".", ";")) and not l.endswith(":"):
                ops.append(l)
        loops = []
        for i, l in enumerate(ops):
            mm = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] <= i:
                loops.append((labels[mm.group(1)], i))
        vg = re.search(r"\.set %s\.num_vgpr, (\d+)" % re.escape(name), text)
        mix = Counter(kind(o.split()[0]) for o in ops)
        print("%s\n  total %d %s vgpr %s" % (name, len(ops), dict(mix), vg.group(1) if vg else "?"))
        for a, b in sorted(loops, key=lambda t: t[0] - t[1])[:3]:
            print("  loop [%d..%d] %d instr %s" % (a, b, b - a + 1, dict(Counter(kind(o.split()[0]) for o in ops[a:b + 1]))))


if __name__ == "__main__":
    main()
