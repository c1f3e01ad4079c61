"""tests/golden/make_golden.py -- regenerate the golden fixtures from the REFERENCE library.

Runs only in the build container (needs /root/reference -> oracle/_ref via oracle/Makefile).
The fixtures hold inputs and the reference's outputs (strict-IEEE build, so they are compiler-flag
independent): golden_quadprog.npz  (daqp_quadprog on hand cases, degenerate cases, config samples)
              golden_warm.npz      (setup_daqp -> solve -> {update_ldp(UPDATE_v) -> solve}*)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle.pin_oracle import edge_cases  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = O.Reference(strict=True)
    out = {}

    def add(name, q):
        x, lam, fval, flag, it = ref.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q.get("sense"))
        A = np.ascontiguousarray(q["A"], dtype=np.float64).reshape(-1, q["f"].size)
        for k, v in dict(H=q["H"], f=q["f"], A=A, bupper=q["bupper"], blower=q["blower"], x=x, lam=lam,
                         fval=np.float64(fval), exitflag=np.int32(flag), iter=np.int32(it)).items():
            out[f"{name}/{k}"] = np.asarray(v)
        if q.get("sense") is not None:
            out[f"{name}/sense"] = np.asarray(q["sense"], np.int32)

    for name, q in edge_cases(np.random.default_rng(7)):
        add("edge_" + name, q)
    for cfg, cnt in (("C1", 8), ("C2", 4), ("C3", 12)):
        n, m, ms, na, seed, _ = O.CONFIGS[cfg]
        for k in range(cnt):
            add(f"{cfg}_{k:02d}", O.generate_qp(n, m, ms, na, rng=[seed, k]))
    for trial in range(24):
        rng = np.random.default_rng([99, trial])
        eps = 10.0 ** rng.uniform(-13, -2)
        n = int(rng.integers(4, 12)); m = int(rng.integers(n + 4, 3 * n)); ms = int(rng.integers(0, min(n, m // 3) + 1))
        na = int(rng.integers(1, min(n, m - ms)))
        add(f"nasty_{trial:02d}", O.generate_nasty(n, m, ms, na, eps, rng, n_dup=int(rng.integers(0, 5)),
                                                   n_eq=int(rng.integers(0, 3)), n_soft=int(rng.integers(0, 3)),
                                                   dep_eq=bool(rng.integers(0, 2))))
    np.savez_compressed(os.path.join(HERE, "golden_quadprog.npz"), **out)

    n, m, ms, na, seed, _ = O.CONFIGS["C1"]
    q = O.generate_qp(n, m, ms, na, rng=[seed, 1])
    rm = ref.model(n, m, ms)
    assert rm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None) == 1
    fs, xs, its, flags = [q["f"].copy()], [], [], []
    for t in range(6):
        if t > 0:
            fs.append(fs[-1] + 0.05 * np.random.default_rng([45, 1, t]).standard_normal(n))
            assert rm.update(O.UPDATE_v, f=fs[-1]) == 0
        x, lam, fval, flag, it = rm.solve()
        xs.append(x); its.append(it); flags.append(flag)
    rm.close()
    np.savez_compressed(os.path.join(HERE, "golden_warm.npz"), n=n, m=m, ms=ms, H=q["H"], f0=q["f"], A=q["A"],
                        bupper=q["bupper"], blower=q["blower"], fs=np.array(fs), x=np.array(xs),
                        iter=np.array(its, np.int32), exitflag=np.array(flags, np.int32))
    print("wrote", len({k.split('/')[0] for k in out}), "quadprog cases; warm iters", its)


if __name__ == "__main__":
    main()
