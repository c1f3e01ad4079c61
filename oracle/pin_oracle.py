"""oracle/pin_oracle.py -- pins the C restatement against the reference library itself.

Runs in the build container (needs oracle/_ref, i.e. /root/reference).  For every
case it demands BIT-identical x, lam, fval and identical exitflag/iter between
oracle/liboracle.so and the strict-IEEE reference build, and reports the
agreement of the reference's own release (fast-math) build with both.

    python oracle/pin_oracle.py [--n-per-config 300]
"""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same(a, b):
    return np.array_equal(bits(a), bits(b))


def edge_cases(rng):
    """Small hand-made problems exercising the non-random branches."""
    cases = []
    H2, f2 = np.eye(2), np.array([1.0, 1.0])
    A2 = np.array([[1.0, 1.0], [1.0, -1.0]])
    # reference python demo (interfaces/daqp-python/test/example_test.py:17-26)
    cases.append(("py_demo", dict(H=H2, f=f2, A=A2, bupper=np.array([1., 2, 3, 4]), blower=-np.array([1., 2, 3, 4]),
                                  sense=np.zeros(4, np.int32))))
    # _make_qp of example_test.py:175-183 -> x = [-1,-1]
    cases.append(("py_model", dict(H=np.eye(2), f=np.array([2.0, 2.0]), A=np.zeros((0, 2)),
                                   bupper=np.ones(2), blower=-np.ones(2), sense=np.zeros(2, np.int32))))
    # dense H, unconstrained optimum feasible
    q = O.generate_qp(6, 12, 2, 2, rng=rng)
    q2 = dict(q); q2["bupper"] = q["bupper"] + 1e3; q2["blower"] = q["blower"] - 1e3
    cases.append(("unconstrained", q2))
    # crossed bounds -> -1
    q3 = dict(q); q3["bupper"] = q["bupper"].copy(); q3["bupper"][3] = q3["blower"][3] - 1.0
    cases.append(("crossed", q3))
    # infeasible: two parallel constraints that exclude each other
    A = np.array([[1.0, 0.5, 0.0], [1.0, 0.5, 0.0], [0.0, 1.0, 1.0]])
    cases.append(("infeasible", dict(H=np.array([[2.0, 0.3, 0], [0.3, 1.0, 0.1], [0, 0.1, 1.5]]), f=np.ones(3), A=A,
                                     bupper=np.array([1.0, 5.0, 2.0]), blower=np.array([-1.0, 3.0, -2.0]),
                                     sense=np.zeros(3, np.int32))))
    # equality (sense 5) + soft (sense 8) + a pre-activated lower bound (sense 3)
    q4 = O.generate_qp(8, 20, 3, 4, rng=rng)
    s = np.zeros(20, np.int32); s[5] = 5; s[7] = 8; s[9] = 8; s[11] = 3
    bu, bl = q4["bupper"].copy(), q4["blower"].copy()
    bl[5] = bu[5]
    bu[7] = bl[7] + 1e-3
    q4.update(sense=s, bupper=bu, blower=bl)
    cases.append(("eq_soft_warm", q4))
    # unmarked equality (bupper == blower) found by check_bounds
    q5 = O.generate_qp(8, 20, 0, 4, rng=rng)
    bu, bl = q5["bupper"].copy(), q5["blower"].copy(); bl[2] = bu[2]; bl[13] = bu[13]
    q5.update(bupper=bu, blower=bl)
    cases.append(("implicit_eq", q5))
    # zero row in A (normalize_M marks it immutable)
    q6 = O.generate_qp(6, 14, 0, 3, rng=rng)
    A6 = q6["A"].copy(); A6[4] = 0; bu = q6["bupper"].copy(); bl = q6["blower"].copy(); bu[4] = 1; bl[4] = -1
    q6.update(A=A6, bupper=bu, blower=bl)
    cases.append(("zero_row", q6))
    # zero row with excluding bounds -> infeasible at setup
    q7 = dict(q6); bu7 = bu.copy(); bl7 = bl.copy(); bu7[4] = -1.0; bl7[4] = -2.0
    q7.update(bupper=bu7, blower=bl7)
    cases.append(("zero_row_infeasible", q7))
    # nearly dependent constraints: exercises pivoting / singular branches / refinement
    for t, eps in enumerate([1e-4, 1e-7, 1e-9, 1e-12]):
        qq = O.generate_qp(10, 30, 0, 6, rng=rng)
        A = qq["A"].copy()
        A[1] = A[0] + eps * rng.standard_normal(10)
        A[3] = A[2] * (1 + eps) + eps * rng.standard_normal(10)
        A[5] = A[0] + A[2] + eps * rng.standard_normal(10)
        bu = qq["bupper"].copy(); bl = qq["blower"].copy()
        bu[:6] = -np.abs(bu[:6]) * 0.1 - 0.5          # force violations on the dependent rows
        bl[:6] = bu[:6] - 1.0
        qq.update(A=A, bupper=bu, blower=bl)
        cases.append((f"near_dep_{t}", qq))
    # more active constraints than variables possible (m >> n, tight box)
    qq = O.generate_qp(4, 30, 4, 3, rng=rng)
    qq.update(bupper=qq["bupper"] - 0.3, blower=qq["blower"] + 0.0)
    cases.append(("tight", qq))
    # diagonal (non-identity) Hessian with simple bounds
    qd = O.generate_qp(7, 18, 5, 4, rng=rng)
    qd.update(H=np.diag(1.0 + 3 * rng.random(7)))
    cases.append(("diag_H", qd))
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-per-config", type=int, default=300)
    args = ap.parse_args()
    ora, strict, fast = O.Oracle(), O.Reference(strict=True), O.Reference(strict=False)
    bad = 0
    total = 0

    def check(name, q, settings=None):
        nonlocal bad, total
        a = ora.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q.get("sense"), settings)
        b = strict.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q.get("sense"), settings)
        total += 1
        ok = a[3] == b[3] and (a[3] < 0 and a[4] == b[4] or
                               (a[4] == b[4] and same(a[0], b[0]) and same(a[1], b[1]) and same(a[2], b[2])))
        if a[3] < 0 and b[3] < 0 and a[3] == b[3]:
            ok = a[3] == b[3] and (a[4] == b[4] or b[4] == 0)
        if not ok:
            bad += 1
            print(f"MISMATCH {name}: oracle flag/iter {a[3]}/{a[4]} ref {b[3]}/{b[4]} "
                  f"dx {np.abs(a[0] - b[0]).max():.3e}")
        return a, b

    rng = np.random.default_rng(7)
    for name, q in edge_cases(rng):
        a, b = check(name, q)
        print(f"  {name:22s} flag {b[3]:3d} iter {b[4]:4d}")
    # iteration limit (core_tests.jl:33-35)
    q = O.generate_qp(20, 40, 0, 8, rng=[1234, 0])
    a, b = check("iter_limit_1", q, O.default_settings(iter_limit=1))
    print(f"  iter_limit=1           flag {b[3]} iter {b[4]}")

    # the two branches of daqp_ldp that ordinary data never reaches, forced through settings (tests/test_gpu_branches.py):
    # refactor repair (daqp.c:33-46) and cycle guard (daqp.c:66-85, exit flag -2)
    forced = {"refactor": dict(refactor_tol=10.0), "cycle": dict(progress_tol=1e30, cycle_tol=0)}
    for (n, m, ms, na), N in (((20, 40, 0, 8), 20), ((12, 48, 12, 6), 40), ((9, 30, 4, 4), 24), ((24, 60, 6, 8), 8), ((50, 150, 0, 20), 6), ((70, 160, 5, 25), 8)):
        qs = O.generate_batch(N, n, m, ms, na, 4242 + n, start=100)
        for name, kw in forced.items():
            flags = set()
            for k in range(N):
                q = {key: qs[key][k] for key in ("H", "f", "A", "bupper", "blower")}
                a, b = check(f"forced_{name}[{n},{k}]", q, O.default_settings(**kw))
                flags.add(b[3])
            print(f"  forced {name:9s} n={n:3d} m={m:3d}: {N} QPs, exit flags {sorted(flags)}")

    for cfg, (n, m, ms, na, seed, _) in O.CONFIGS.items():
        N = args.n_per_config if n < 100 else max(10, args.n_per_config // 15)
        nfast_same = 0
        dxmax = 0.0
        its = []
        for k in range(N):
            q = O.generate_qp(n, m, ms, na, rng=[seed, k])
            a, b = check(f"{cfg}[{k}]", q)
            c = fast.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
            nfast_same += int(c[4] == b[4] and c[3] == b[3] and np.array_equal(np.sign(c[1]), np.sign(b[1])))
            dxmax = max(dxmax, np.abs(c[0] - b[0]).max(), np.abs(b[0] - q["x"]).max())
            its.append(b[4])
        print(f"{cfg}: n={n} m={m} ms={ms}: {N} QPs, mean iter {np.mean(its):.1f} max {max(its)}; "
              f"release-flags reference: identical active set+iter on {nfast_same}/{N}, max|dx| {dxmax:.2e}")

    # workspace sequence: setup_daqp -> solve -> {update(v) -> solve}* 
    n, m, ms, na, seed, _ = O.CONFIGS["C2"]
    for k in range(max(3, args.n_per_config // 30)):
        q = O.generate_qp(n, m, ms, na, rng=[seed, k])
        om, rm = ora.model(n, m, ms), strict.model(n, m, ms)
        fa, fb = om.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"]), \
            rm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        seq = [(om.solve(), rm.solve())]
        f = q["f"].copy()
        for t in range(5):
            f = f + 0.05 * np.random.default_rng([45, k, t]).standard_normal(n)
            ua, ub = om.update(O.UPDATE_v, f=f), rm.update(O.UPDATE_v, f=f)
            assert ua == ub == 0, (ua, ub)
            seq.append((om.solve(), rm.solve()))
        for t, (a, b) in enumerate(seq):
            total += 1
            if not (a[3] == b[3] and a[4] == b[4] and same(a[0], b[0]) and same(a[1], b[1]) and same(a[2], b[2])):
                bad += 1
                print(f"MISMATCH warm[{k}][{t}] {a[3]}/{a[4]} vs {b[3]}/{b[4]}")
        if k == 0:
            print("warm sequence iters:", [b[4] for _, b in seq], "setup flags", fa, fb)
        rm.close()
    # proximal outer loop (daqp_prox.c; utils.c:223-432): semidefinite / diagonal-with-zeros / forcibly shifted Hessians
    nprox = args.n_per_config
    flags, outer_its = {}, []
    for k in range(nprox):
        rng = np.random.default_rng([201, k])
        n = int(rng.integers(2, 30)); m = int(rng.integers(n, 3 * n + 2)); ms = int(rng.integers(0, min(n, m) + 1)) if k % 3 == 0 else 0
        q = O.generate_singular_qp(n, m, ms, rank=int(rng.integers(1, n + 1)), rng=[202, k], kind="diag" if k % 4 == 1 else "dense",
                                   in_range=(k % 2 == 0))
        st = None
        if k % 4 == 2:
            st = O.default_settings(eps_prox=10.0 ** rng.uniform(-5, 0), eta_prox=(-1.0 if k % 8 == 2 else 10.0 ** rng.uniform(-10, -6)))
        if k % 16 == 7:
            q["H"] = q["H"] - 2.0 * np.eye(n)
        if k % 16 == 11:
            st = O.default_settings(iter_limit=int(rng.integers(2, 30)))
        a, b = check(f"prox[{k}]", q, st)
        flags[b[3]] = flags.get(b[3], 0) + 1
        outer_its.append(b[4])
    print(f"proximal: {nprox} QPs, exit flags {flags}, mean total iterations {np.mean(outer_its):.1f} max {max(outer_its)}")
    for k in range(max(3, args.n_per_config // 10)):
        rng = np.random.default_rng([203, k])
        n = int(rng.integers(4, 30)); m = int(rng.integers(n, 3 * n + 2)); ms = 0 if k % 2 else min(3, n)
        q = O.generate_singular_qp(n, m, ms, rank=int(rng.integers(1, n)), rng=[204, k], kind="diag" if k % 5 == 0 else "dense",
                                   in_range=(k % 3 == 0))
        st = O.default_settings(eps_prox=1e-2, eta_prox=1e-9) if k % 2 else None
        om, rm = ora.model(n, m, ms, settings=st), strict.model(n, m, ms, settings=st)
        fa, fb = om.setup(**q), rm.setup(**q)
        assert fa == fb, (fa, fb)
        f = q["f"].copy()
        for t in range(4):
            a, b = om.solve(), rm.solve()
            total += 1
            if not (a[3] == b[3] and a[4] == b[4] and same(a[0], b[0]) and same(a[1], b[1]) and same(a[2], b[2])):
                bad += 1
                print(f"MISMATCH prox warm[{k}][{t}] {a[3]}/{a[4]} vs {b[3]}/{b[4]}")
            f = f + 0.3 * rng.standard_normal(n)
            assert om.update(O.UPDATE_v, f=f) == rm.update(O.UPDATE_v, f=f) == 0
        rm.close()
    # linear programs (H == NULL): the LP branch of daqp_prox.c -- R = I, adaptive smoothing, gradient steps, unbounded rays
    flags, its = {}, []
    for k in range(args.n_per_config):
        rng = np.random.default_rng([401, k])
        n = int(rng.integers(2, 25)); m = int(rng.integers(n + 1, 3 * n + 3)); ms = int(rng.integers(0, min(n, m) + 1)) if k % 2 else 0
        q = O.generate_lp(n, m, ms, [402, k], unbounded=(k % 7 == 3))
        st = None
        if k % 5 == 4:
            st = O.default_settings(eta_prox=1e-9)
        if k % 11 == 5:
            st = O.default_settings(iter_limit=int(rng.integers(2, 15)))
        a, b = check(f"lp[{k}]", q, st)
        flags[b[3]] = flags.get(b[3], 0) + 1
        its.append(b[4])
    print(f"linear programs: {args.n_per_config} LPs, exit flags {flags}, mean total iterations {np.mean(its):.1f} max {max(its)}")
    for k in range(max(3, args.n_per_config // 10)):
        rng = np.random.default_rng([403, k])
        n = int(rng.integers(3, 25)); m = int(rng.integers(n + 1, 3 * n + 3)); ms = 0 if k % 2 else min(3, n)
        q = O.generate_lp(n, m, ms, [404, k])
        om, rm = ora.model(n, m, ms), strict.model(n, m, ms)
        assert om.setup(**q) == rm.setup(**q) == 1
        f = q["f"].copy()
        for t in range(4):
            a, b = om.solve(), rm.solve()
            total += 1
            if not (a[3] == b[3] and a[4] == b[4] and (a[3] < 0 or (same(a[0], b[0]) and same(a[1], b[1]) and same(a[2], b[2])))):
                bad += 1
                print(f"MISMATCH lp warm[{k}][{t}] {a[3]}/{a[4]} vs {b[3]}/{b[4]}")
            f = f + 0.3 * rng.standard_normal(n)
            assert om.update(O.UPDATE_v, f=f) == rm.update(O.UPDATE_v, f=f) == 0
        rm.close()
    # equalities and soft rows inside the proximal loop (QPs and LPs)
    for k in range(args.n_per_config):
        rng = np.random.default_rng([601, k])
        n = int(rng.integers(3, 25)); m = int(rng.integers(n + 3, 3 * n + 4)); ms = int(rng.integers(0, min(n, 4) + 1)) if k % 2 else 0
        if k % 3 == 0:
            q = O.generate_lp(n, m, ms, [602, k])
        else:
            q = O.generate_singular_qp(n, m, ms, rank=int(rng.integers(1, n)), rng=[603, k], kind="diag" if k % 4 == 1 else "dense", in_range=(k % 5 == 0))
        q = O.add_sense_variety(q, ms, int(rng.integers(0, min(4, n - 1) + 1)), int(rng.integers(0, 3)), [604, k])
        check(f"prox_sense[{k}]", q, O.default_settings(eps_prox=1e-2, eta_prox=1e-8) if k % 5 == 0 else None)
    # daqp_set_primal_start (api.c:636-641): the proximal loop from a given iterate
    for k in range(max(10, args.n_per_config // 3)):
        rng = np.random.default_rng([701, k])
        n = int(rng.integers(3, 25)); m = int(rng.integers(n + 2, 3 * n + 3)); ms = 0 if k % 2 else min(2, n)
        q = O.generate_lp(n, m, ms, [702, k]) if k % 3 == 0 else O.generate_singular_qp(n, m, ms, rank=int(rng.integers(1, n)), rng=[703, k], in_range=(k % 2 == 0))
        st = O.default_settings(eps_prox=1e-2, eta_prox=1e-8) if k % 4 == 0 else None
        om, rm = ora.model(n, m, ms, settings=st), strict.model(n, m, ms, settings=st)
        assert om.setup(**q) == rm.setup(**q) == 1
        x0 = rng.standard_normal(n)
        om.set_primal_start(x0); rm.set_primal_start(x0)
        a, b = om.solve(), rm.solve()
        total += 1
        if not (a[3] == b[3] and a[4] == b[4] and (a[3] < 0 or (same(a[0], b[0]) and same(a[1], b[1]) and same(a[2], b[2])))):
            bad += 1
            print(f"MISMATCH primal_start[{k}] {a[3]}/{a[4]} vs {b[3]}/{b[4]}")
        rm.close()
    # the MPC usage with a semidefinite plant Hessian (tests/test_gpu_prox.py::test_shared_singular_hessian): setup_daqp with
    # open bounds, daqp_update_ldp(UPDATE_v|UPDATE_d) with the problem's own f / bounds, daqp_solve (the proximal loop), a warm re-solve
    for k in range(max(8, args.n_per_config // 5)):
        rng = np.random.default_rng([801, k])
        n = int(rng.integers(4, 40)); m = int(rng.integers(n + 2, 3 * n + 3)); ms = 0 if k % 2 else min(4, n)
        q = O.generate_singular_qp(n, m, ms, rank=max(1, n // 2), rng=[802, k], kind="diag" if k % 3 == 1 else "dense")
        om, rm = ora.model(n, m, ms), strict.model(n, m, ms)
        wide_u, wide_l = np.full(m, 1e30), np.full(m, -1e30)
        assert om.setup(q["H"], q["f"], q["A"], wide_u, wide_l, None) == rm.setup(q["H"], q["f"], q["A"], wide_u, wide_l, None)
        f = q["f"] + 0.2 * rng.standard_normal(n)
        kw = dict(f=f, bupper=q["bupper"], blower=q["blower"])
        assert om.update(O.UPDATE_v | O.UPDATE_d, **kw) == rm.update(O.UPDATE_v | O.UPDATE_d, **kw) == 0
        for t in range(2):
            a, b = om.solve(), rm.solve()
            total += 1
            if not (a[3] == b[3] and a[4] == b[4] and (a[3] < 0 or (same(a[0], b[0]) and same(a[1], b[1]) and same(a[2], b[2])))):
                bad += 1
                print(f"MISMATCH shared_singular[{k}][{t}] {a[3]}/{a[4]} vs {b[3]}/{b[4]}")
            f = f + 0.05 * rng.standard_normal(n)
            assert om.update(O.UPDATE_v, f=f) == rm.update(O.UPDATE_v, f=f) == 0
        rm.close()
    # every daqp_update_ldp mask (utils.c:58-221 runs each bit's step on its own; daqp.pyx:513-571 builds them field by field): 31 masks
    # x 4 shapes x sense variety x 3 consecutive updates, incl. a sense bit without a sense array (utils.c:85-86), unmarked equalities
    # that the bound check marks, and crossed bounds (the update ends with -1 and the next solve runs on what the workspace held)
    R_, M_, V_, D_, S_ = O.UPDATE_Rinv, O.UPDATE_M, O.UPDATE_v, O.UPDATE_d, O.UPDATE_sense
    mcount = 0
    for (n, m, ms, na) in ((20, 40, 0, 8), (12, 48, 12, 6), (10, 30, 4, 5), (30, 70, 10, 9)):
        for trial in range(max(2, args.n_per_config // 100)):
            q = O.generate_qp(n, m, ms, na, rng=[7, n, trial])
            H, f, A, bu, bl = q["H"], q["f"], q["A"], q["bupper"], q["blower"]
            sense = np.zeros(m, np.int32)
            if trial % 3 == 1:
                sense[ms + 1] = 5; bl[ms + 1] = bu[ms + 1]; sense[ms + 3] = 8
            if trial % 3 == 2:
                sense[m - 1] = 8; sense[0] = 8 if ms else 0
            for mask in range(1, 32):
                om, rm = ora.model(n, m, ms, ns=int((sense & 8).sum()) + 2), strict.model(n, m, ms)
                assert om.setup(H, f, A, bu, bl, sense) == rm.setup(H, f, A, bu, bl, sense)
                rr = (om.solve(), rm.solve())[1]
                for step in range(3):
                    r2 = np.random.default_rng([9, n, trial, mask, step])
                    kw = {}
                    if mask & R_:
                        G = r2.standard_normal((n, n)) * 0.1; kw["H"] = H + G @ G.T
                    if mask & M_: kw["A"] = A + 0.05 * r2.standard_normal(A.shape)
                    if mask & V_: kw["f"] = f + 0.3 * r2.standard_normal(n)
                    if mask & D_:
                        w = 0.05 * r2.random(m); kw["bupper"] = bu + w; kw["blower"] = bl - 0.5 * w
                        if trial % 3 == 1: kw["blower"][ms + 1] = kw["bupper"][ms + 1]
                        if step == 1 and trial % 2 == 0: kw["blower"][ms + 4] = kw["bupper"][ms + 4]          # unmarked equality
                        if step == 0 and trial % 4 == 3: kw["bupper"][ms + 6] = kw["blower"][ms + 6] - 1.0    # crossed
                    if mask & S_:
                        s2 = sense.copy()
                        act = np.nonzero(rr[1])[0]
                        if step == 0 and len(act): s2[act[0]] |= 1 | (2 if rr[1][act[0]] < 0 else 0)
                        if step == 1: s2[r2.integers(0, m)] |= 1
                        kw["sense"] = s2 if not (step == 2 and trial % 2) else None
                        if kw["sense"] is None:     # qp->sense == NULL with the sense bit: zeros, nothing rebuilt
                            om.keep["sense"] = None; rm.keep["sense"] = None
                    uo, ur = om.update(mask, **kw), rm.update(mask, **kw)
                    ro, rr = om.solve(), rm.solve()
                    total += 1; mcount += 1
                    ok = uo == ur and ro[3] == rr[3] and ro[4] == rr[4] and (ro[3] < 0 or (same(ro[0], rr[0]) and same(ro[1], rr[1]) and same(ro[2], rr[2])))
                    ok = ok and np.array_equal(om.state()[0], rm.working_set())
                    if not ok:
                        bad += 1
                        print(f"MISMATCH update_mask n={n} trial {trial} mask {mask} step {step}: update {uo}/{ur} solve {ro[3]}/{ro[4]} vs {rr[3]}/{rr[4]}")
                rm.close()
    print(f"  update masks: {mcount} update + solve steps")
    print(f"pin result: {total - bad}/{total} bit-identical to the strict reference build")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
