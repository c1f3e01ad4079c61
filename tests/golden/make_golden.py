"""tests/golden/make_golden.py -- regenerate the golden fixtures from the REFERENCE library.

Runs only in the build container (needs /root/reference -> oracle/_ref via oracle/Makefile).
The fixtures hold inputs and the reference's outputs (strict-IEEE build, so they are compiler-flag
independent): golden_quadprog.npz  (daqp_quadprog on hand cases, degenerate cases, config samples)
              golden_warm.npz      (setup_daqp -> solve -> {update_ldp(UPDATE_v) -> solve}*)
              golden_eliminated.npz (daqp_quadprog on equality-heavy QPs, which the reference pre-reduces: eq_elim.c)
              golden_prox.npz      (singular / forcibly shifted Hessians: daqp_quadprog through daqp_prox, and a
                                    setup_daqp -> solve -> {update_ldp(UPDATE_v) -> solve}* sequence on one)
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

    # ---- proximal outer loop (daqp_prox.c): positive SEMI-definite Hessians and eps_prox > 0
    px = {}
    cases = []
    for k in range(40):
        rng = np.random.default_rng([101, k])
        n = int(rng.integers(3, 26)); m = int(rng.integers(n + 2, 3 * n + 2)); ms = int(rng.integers(0, min(n, 5) + 1)) if k % 3 == 0 else 0
        kind = "diag" if k % 4 == 3 else "dense"
        q = O.generate_singular_qp(n, m, ms, rank=int(rng.integers(1, n)), rng=[102, k], kind=kind, in_range=(k % 2 == 1))
        kw = {}
        if k % 5 == 1:
            kw = dict(eps_prox=10.0 ** rng.uniform(-4, -1))
        if k % 5 == 2:
            kw = dict(eps_prox=-1e-2, eta_prox=1e-9)
        if k == 13:
            kw = dict(iter_limit=9)
        if k == 17:
            q["H"] = q["H"] - 1e4 * np.eye(n)   # indefinite beyond 16 doublings of the shift: -5
        cases.append((f"prox_{k:02d}", q, kw))
    qd = O.generate_qp(9, 20, 2, 4, rng=[103, 0])
    cases.append(("prox_forced_definite", {kk: qd[kk] for kk in ("H", "f", "A", "bupper", "blower", "sense")}, dict(eps_prox=1e-3)))
    for k in range(24):   # linear programs: H is None (stored as an empty array)
        rng = np.random.default_rng([111, k])
        n = int(rng.integers(2, 22)); m = int(rng.integers(n + 1, 3 * n + 3)); ms = int(rng.integers(0, min(n, m) + 1)) if k % 2 else 0
        kw = dict(iter_limit=11) if k == 7 else (dict(eta_prox=1e-9) if k % 5 == 4 else {})
        cases.append((f"lp_{k:02d}", O.generate_lp(n, m, ms, [112, k], unbounded=(k % 6 == 3)), kw))
    for name, q, kw in cases:
        st = O.default_settings(**kw)
        x, lam, fval, flag, it = ref.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"], settings=st)
        for kk, v in dict(H=(q["H"] if q["H"] is not None else np.zeros((0, 0))), f=q["f"], A=np.ascontiguousarray(q["A"]).reshape(-1, q["f"].size),
                          bupper=q["bupper"], blower=q["blower"], sense=q["sense"], x=x, lam=lam, fval=np.float64(fval), exitflag=np.int32(flag), iter=np.int32(it),
                          eps_prox=np.float64(kw.get("eps_prox", -1e-6)), eta_prox=np.float64(kw.get("eta_prox", -1.0)),
                          iter_limit=np.int32(kw.get("iter_limit", 10000))).items():
            px[f"{name}/{kk}"] = np.asarray(v)
    n, m, ms = 12, 30, 3
    q = O.generate_singular_qp(n, m, ms, rank=5, rng=[104, 0])
    rm = ref.model(n, m, ms)
    assert rm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None) == 1
    fs, xs, lams, fvs, its, flags = [q["f"].copy()], [], [], [], [], []
    for t in range(5):
        if t > 0:
            fs.append(fs[-1] + 0.4 * np.random.default_rng([105, t]).standard_normal(n))
            assert rm.update(O.UPDATE_v, f=fs[-1]) == 0
        x, lam, fval, flag, it = rm.solve()
        xs.append(x); lams.append(lam); fvs.append(fval); its.append(it); flags.append(flag)
    rm.close()
    for kk, v in dict(n=n, m=m, ms=ms, H=q["H"], A=q["A"], bupper=q["bupper"], blower=q["blower"], fs=np.array(fs), x=np.array(xs),
                      lam=np.array(lams), fval=np.array(fvs), iter=np.array(its, np.int32), exitflag=np.array(flags, np.int32)).items():
        px[f"warm/{kk}"] = np.asarray(v)
    # ---- daqp_quadprog on equality-heavy QPs: the reference eliminates the equalities first (eq_elim.c); this library
    # solves the full LDP instead, so these fixtures are compared with a tolerance (and not by the oracle, which reports
    # the unbuilt reduction as -8)
    el = {}
    for k in range(16):
        rng = np.random.default_rng([121, k])
        n = int(rng.integers(8, 40)); ms = int(rng.integers(0, 5)) if k % 2 else 0
        neq = int(rng.integers(max(6, n // 10 + 1), max(7, n // 2)))
        m = ms + neq + int(rng.integers(n, 2 * n))
        q = O.generate_equality_qp(n, m, ms, neq, [122, k])
        x, lam, fval, flag, it = ref.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        for kk, v in dict(H=q["H"], f=q["f"], A=q["A"], bupper=q["bupper"], blower=q["blower"], sense=q["sense"], x=x, lam=lam,
                          fval=np.float64(fval), exitflag=np.int32(flag), iter=np.int32(it)).items():
            el[f"elim_{k:02d}/{kk}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "golden_eliminated.npz"), **el)
    print("wrote 16 equality-eliminated cases: flags", sorted({int(el[f'elim_{k:02d}/exitflag']) for k in range(16)}))
    np.savez_compressed(os.path.join(HERE, "golden_prox.npz"), **px)
    print("wrote", len(cases), "proximal cases: flags", sorted({int(px[f'{c[0]}/exitflag']) for c in cases}), "warm iters", its)


if __name__ == "__main__":
    main()
