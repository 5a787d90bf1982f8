"""Round-6 probe for the flaky `teacher_diff` of ONE rank in the 8-rank FCN sharded test: runs the test's rank processes with
`SAC._momentum_update` wrapped so that every call also evaluates sum_t ||teacher_t - student_t||_2 with plain torch ops at the same
point of the stream, plus a checksum of student and teacher.  Usage: python tools/experiments/r6_teacher_diff_probe.py [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "da-sac_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def rank_main(rank, port, case, q):
    for p in (ROOT, os.path.join(ROOT, "da-sac_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    from models import sac as sac_mod
    real = sac_mod.SAC._momentum_update
    log = []

    def wrapped(self, update=False):
        out = real(self, update)
        if self._ema_plan is None:
            return out
        plan = self._ema_plan[1]
        fast, slow = self._ema_pairs()
        with torch.no_grad():          # everything stays on the stream: no host read here (the first version's float() hid the flake)
            per_kernel = plan.sq[:plan.n_tensors].clone()
            per_torch = torch.stack([(s_.double() - f.double()).pow(2).sum() for f, s_ in zip(fast, slow)])
        log.append((bool(update), out, per_kernel, per_torch))
        return out
    sac_mod.SAC._momentum_update = wrapped
    import test_gpu_sharded as T

    class Q:
        def put(self, item):
            names = [k for k in net_keys(sac_mod) ]
            rows = []
            for upd, out, pk, pt in log:
                pk, pt = pk.cpu(), pt.cpu()
                bad = [(i, float(pk[i]), float(pt[i])) for i in range(pk.numel()) if abs(float(pk[i]) - float(pt[i])) > 1e-6 * max(float(pt[i]), 1e-30)]
                rows.append((upd, float(out), float(pk.sqrt().sum()), float(pt.sqrt().sum()), bad[:6]))
            q.put((item[0], [r["teacher_diff"] for r in item[1]], rows))
    T._rank_main(rank, port, case, Q())


def net_keys(_):
    return []


if __name__ == "__main__":
    from conftest import run_ranks
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    case = ("fcn_vgg16_bn", 8, 2, 4)
    for rep in range(reps):
        got = run_ranks(rank_main, 8, lambda r, port, q: (r, port, case, q), timeout=300)
        for r, tds, rows in got:
            flag = [(i, row) for i, row in enumerate(rows) if row[4] or abs(row[1] - row[3]) > 1e-5 * max(row[3], 1e-9)]
            print("rep", rep, "rank", r, "teacher_diff per iteration", tds, "calls (kernel out, sum sqrt kernel sq, sum sqrt torch sq)",
                  [(round(a, 6), round(b, 6), round(c, 6)) for _, a, b, c, _ in rows], "MISMATCH " + str(flag) if flag else "")
