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
        fast, slow = self._ema_pairs()
        with torch.no_grad():
            ref = sum(float((s.double() - f.double()).norm()) for f, s in zip(fast, slow))
            cs = (float(sum(f.double().sum() for f in fast)), float(sum(s.double().sum() for s in slow)))
        log.append((bool(update), float(out), ref, cs))
        return out
    sac_mod.SAC._momentum_update = wrapped
    import test_gpu_sharded as T

    class Q:
        def put(self, item):
            q.put((item[0], [(r["teacher_diff"]) for r in item[1]], log))
    T._rank_main(rank, port, case, Q())


if __name__ == "__main__":
    from conftest import run_ranks
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    case = ("fcn_vgg16_bn", 8, 2, 4)
    for rep in range(reps):
        got = run_ranks(rank_main, 8, lambda r, port, q: (r, port, case, q), timeout=300)
        ref = got[0][2]
        for r, tds, log in got:
            flag = ""
            for i, (upd, k, t, cs) in enumerate(log):
                if abs(k - t) > 1e-4 * max(abs(t), 1e-9) or cs != ref[i][3]:
                    flag += " [call %d: kernel %.6f torch %.6f checksums %s vs rank0 %s]" % (i, k, t, cs, ref[i][3])
            print("rep", rep, "rank", r, "teacher_diff per iteration", tds, "calls", [(round(k, 6), round(t, 6)) for _, k, t, _ in log], flag)
