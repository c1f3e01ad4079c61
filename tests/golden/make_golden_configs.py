"""tests/golden/make_golden_configs.py -- reference-written fixtures for the two BASELINE configurations that tests/golden lacked:
     golden_c4.npz       daqp_quadprog of the REFERENCE (strict build) on 4 QPs of config C4 (n=200, m=600, 80 active): inputs and x, lam,
                         fval, exitflag, iter.  QP k is oracle.generate_qp(200, 600, 0, 80, rng=[44, k]) with every input array ROUNDED TO
                         fp32-representable values before the reference sees it, and stored as float32: half the bytes (2.6 MB instead of
                         5.1), and the stored problem is exactly the one that was solved.  (The generator itself cannot stand in for the
                         inputs: it multiplies matrices through the host's BLAS, whose rounding differs from CPU to CPU.)
     golden_warm_c2.npz  config C5's shape: setup_daqp -> daqp_solve -> 10 x {daqp_update_ldp(UPDATE_v) -> daqp_solve} of the REFERENCE on
                         2 QPs of config C2 (n=50, m=150), f walking as SURVEY 8(d) says (f += 0.05 N(0,I), default_rng([45, k, t])):
                         inputs in full (the walk included) and x, lam, fval, iter, exitflag of every step.
Runs only in the build container (needs /root/reference -> oracle/_ref via oracle/Makefile), like make_golden.py.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("H", "f", "A", "bupper", "blower")


def main():
    ref = O.Reference(strict=True)
    n, m, ms, na, seed, _ = O.CONFIGS["C4"]
    out = dict(n=n, m=m, ms=ms, n_active=na, seed=seed)
    for k in range(4):
        q = O.generate_qp(n, m, ms, na, rng=[seed, k])
        for kk in KEYS:
            out[f"{k}/{kk}"] = np.ascontiguousarray(q[kk], np.float32)
            q[kk] = out[f"{k}/{kk}"].astype(np.float64)
        x, lam, fval, flag, it = ref.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
        for kk, v in dict(x=x, lam=lam, fval=np.float64(fval), exitflag=np.int32(flag), iter=np.int32(it)).items():
            out[f"{k}/{kk}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "golden_c4.npz"), **out)
    print("C4:", [(int(out[f"{k}/exitflag"]), int(out[f"{k}/iter"])) for k in range(4)])

    n, m, ms, na, seed, _ = O.CONFIGS["C2"]
    T = 10
    w = dict(n=n, m=m, ms=ms, T=T)
    for k in range(2):
        q = O.generate_qp(n, m, ms, na, rng=[seed, k])
        rm = ref.model(n, m, ms)
        assert rm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None) == 1
        fs, xs, lams, fvs, its, flags = [q["f"].copy()], [], [], [], [], []
        for t in range(T + 1):
            if t > 0:
                fs.append(fs[-1] + 0.05 * np.random.default_rng([45, k, t]).standard_normal(n))
                assert rm.update(O.UPDATE_v, f=fs[-1]) == 0
            x, lam, fval, flag, it = rm.solve()
            xs.append(x); lams.append(lam); fvs.append(fval); its.append(it); flags.append(flag)
        rm.close()
        for kk, v in dict(H=q["H"], A=q["A"], bupper=q["bupper"], blower=q["blower"], fs=np.array(fs), x=np.array(xs), lam=np.array(lams),
                          fval=np.array(fvs), iter=np.array(its, np.int32), exitflag=np.array(flags, np.int32)).items():
            w[f"{k}/{kk}"] = np.asarray(v)
        print("C2 warm", k, "iters", its)
    np.savez_compressed(os.path.join(HERE, "golden_warm_c2.npz"), **w)


if __name__ == "__main__":
    main()
