"""Singular Hessians: the regularising setup passes (reference utils.c:223-432) and the proximal outer loop
(daqp_prox.c:21-221) on the HIP path, against the oracle (itself pinned bit for bit on these cases against the
reference: oracle/pin_oracle.py, tests/golden/prox_*.npz).  The inner solves are the ordinary kernels, so in exact
mode everything is bitwise; in the default (MFMA) mode the shifted problems still are -- their setup always takes the
reference-order kernel -- and the ordinary ones of a mixed batch keep their usual tolerance."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float64).view(np.uint64), np.ascontiguousarray(b, np.float64).view(np.uint64))


def stack(qs):
    return {k: np.stack([q[k] for q in qs]) for k in ("H", "f", "A", "bupper", "blower", "sense")}


def oracle_each(oracle, qs, settings=None):
    return [oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"], settings=settings) for q in qs]


SHAPES = [(6, 14, 0), (12, 30, 4), (20, 60, 0), (33, 70, 5), (50, 150, 0), (70, 150, 6), (64, 128, 0)]   # (last: the register kernel with the hand-over at 65 rows)


@pytest.mark.parametrize("exact", ["1", "0"])
@pytest.mark.parametrize("n,m,ms", SHAPES)
def test_singular_hessians_bitwise(oracle, gpu_lib, monkeypatch, n, m, ms, exact):
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", exact)
    N = 24
    qs = [O.generate_singular_qp(n, m, ms, rank=1 + (k * 7) % (n - 1), rng=[77, n, k], kind="diag" if k % 4 == 3 else "dense")
          for k in range(N)]
    ref = oracle_each(oracle, qs)
    b = stack(qs)
    r = daqp_amd.solve_batch(b["H"], b["f"], b["A"], b["bupper"], b["blower"], b["sense"], ms=ms)
    assert all(rr[3] == 1 for rr in ref)
    for k in range(N):
        x, lam, fval, flag, it = ref[k]
        assert r["exitflag"][k] == flag and r["iter"][k] == it, (k, r["exitflag"][k], flag, r["iter"][k], it)
        assert same(r["x"][k], x) and same(r["lam"][k], lam) and same(r["fval"][k], fval), k


def test_many_outer_iterations(oracle, gpu_lib, monkeypatch):
    """f in the range of H: the minimiser is not a vertex, the proximal map creeps -- tens of outer iterations with
    over-relaxed centres and confirmation steps (daqp_prox.c:159-191), problems of one batch stopping at different times."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    for (n, m, ms), kw in (((12, 30, 2), dict(eps_prox=1e-2, eta_prox=1e-9)), ((24, 50, 0), dict(eps_prox=1e-1, eta_prox=1e-10)),
                           ((50, 150, 0), dict(eps_prox=-1e-2, eta_prox=1e-8))):
        N = 40
        qs = [O.generate_singular_qp(n, m, ms, rank=2 + (k * 5) % (n - 3), rng=[85, n, k], kind="diag" if k % 5 == 4 else "dense",
                                     in_range=True) for k in range(N)]
        st = O.default_settings(**kw)
        ref = oracle_each(oracle, qs, settings=st)
        b = stack(qs)
        mdl = daqp_amd.BatchModel(N, n, m, ms, **kw)
        mdl.setup(b["H"], b["f"], b["A"], b["bupper"], b["blower"], b["sense"], init_mask=64)
        r = mdl.solve()
        outer = mdl.prox_info()["outer"]
        assert outer.max() >= 10 and len(set(outer.tolist())) >= 5, outer
        for k in range(N):
            x, lam, fval, flag, it = ref[k]
            assert r["exitflag"][k] == flag and r["iter"][k] == it, (k, r["exitflag"][k], flag, r["iter"][k], it)
            if flag > 0:
                assert same(r["x"][k], x) and same(r["lam"][k], lam) and same(r["fval"][k], fval), k
        mdl.close()


def test_mixed_batch_and_info(oracle, gpu_lib, monkeypatch):
    """definite and semidefinite Hessians in one batch: the former take one ordinary launch, the latter iterate."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms, N = 16, 40, 3, 32
    qs = []
    for k in range(N):
        if k % 3 == 0:
            q = O.generate_qp(n, m, ms, 5, rng=[78, k])
            qs.append({kk: q[kk] for kk in ("H", "f", "A", "bupper", "blower", "sense")})
        else:
            qs.append(O.generate_singular_qp(n, m, ms, rank=4 + k % 9, rng=[79, k]))
    ref = oracle_each(oracle, qs)
    b = stack(qs)
    mdl = daqp_amd.BatchModel(N, n, m, ms)
    mdl.setup(b["H"], b["f"], b["A"], b["bupper"], b["blower"], b["sense"], init_mask=64)
    r = mdl.solve()
    info = mdl.prox_info()
    for k in range(N):
        x, lam, fval, flag, it = ref[k]
        assert (info["n_prox"][k] > 0) == (k % 3 != 0)
        assert r["exitflag"][k] == flag and r["iter"][k] == it and same(r["x"][k], x) and same(r["lam"][k], lam) and same(r["fval"][k], fval), k
    assert (info["outer"][np.arange(N) % 3 != 0] >= 1).all() and (info["eps"][np.arange(N) % 3 != 0] > 0).all()
    mdl.close()


def test_forced_shift_and_eta(oracle, gpu_lib, monkeypatch):
    """eps_prox > 0 regularises every problem (utils.c:233-281), definite or not; eta_prox sets the stopping rule."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms, N = 10, 24, 2, 16
    qs = []
    for k in range(N):
        if k % 2:
            q = O.generate_qp(n, m, ms, 4, rng=[80, k]); q = {kk: q[kk] for kk in ("H", "f", "A", "bupper", "blower", "sense")}
            if k % 4 == 3:
                q["H"] = np.diag(np.diag(q["H"]))
        else:
            q = O.generate_singular_qp(n, m, ms, rank=3 + k % 5, rng=[81, k])
        qs.append(q)
    for kw in (dict(eps_prox=1e-3), dict(eps_prox=1e-2, eta_prox=1e-8), dict(eps_prox=-1e-4, eta_prox=1e-5)):
        st = O.default_settings(**kw)
        ref = oracle_each(oracle, qs, settings=st)
        b = stack(qs)
        r = daqp_amd.solve_batch(b["H"], b["f"], b["A"], b["bupper"], b["blower"], b["sense"], ms=ms, **kw)
        for k in range(N):
            x, lam, fval, flag, it = ref[k]
            assert r["exitflag"][k] == flag and r["iter"][k] == it, (kw, k, r["exitflag"][k], flag, r["iter"][k], it)
            if flag > 0:
                assert same(r["x"][k], x) and same(r["lam"][k], lam) and same(r["fval"][k], fval), (kw, k)


def test_nonconvex_and_iteration_limit(oracle, gpu_lib, monkeypatch):
    """an indefinite Hessian survives no shift (-5 after 16 doublings or at once with eps_prox = 0); a tiny iteration
    budget ends the outer loop with -4 (daqp_prox.c:201)."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms = 8, 20, 0
    q = O.generate_singular_qp(n, m, ms, rank=4, rng=[82, 0])
    qn = dict(q); qn["H"] = q["H"] - 50.0 * np.eye(n)
    for kw in ({}, dict(eps_prox=0.0), dict(iter_limit=7), dict(iter_limit=2)):
        st = O.default_settings(**kw)
        for prob in (q, qn):
            x, lam, fval, flag, it = oracle.quadprog(prob["H"], prob["f"], prob["A"], prob["bupper"], prob["blower"], prob["sense"], settings=st)
            xg, fg, flg, inf = daqp_amd.solve(prob["H"], prob["f"], prob["A"], prob["bupper"], prob["blower"], prob["sense"], **kw)
            assert flg == flag, (kw, flg, flag)
            if flag != -5:
                assert inf["iterations"] == it, (kw, inf["iterations"], it)
            if flag > 0:
                assert same(xg, x) and fg == fval


def test_model_warm_sequence_and_primal_start(oracle, gpu_lib, monkeypatch):
    """setup once, then solve / update f / solve: the centre of the proximal iterations carries over between solves
    (the reference keeps it in work->x), and daqp_set_primal_start (api.c:636-641) replaces it."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms, N, T = 14, 36, 3, 12, 4
    qs = [O.generate_singular_qp(n, m, ms, rank=3 + k % 8, rng=[83, k], kind="diag" if k % 5 == 4 else "dense") for k in range(N)]
    b = stack(qs)
    oms = [oracle.model(n, m, ms) for _ in range(N)]
    for k, om in enumerate(oms):
        assert om.setup(**qs[k]) == 1
    mdl = daqp_amd.BatchModel(N, n, m, ms)
    mdl.setup(b["H"], b["f"], b["A"], b["bupper"], b["blower"], b["sense"])
    f = b["f"].copy()
    rng = np.random.default_rng(84)
    for t in range(T):
        r = mdl.solve()
        for k, om in enumerate(oms):
            x, lam, fval, flag, it = om.solve()[:5]
            assert r["exitflag"][k] == flag and r["iter"][k] == it, (t, k, r["iter"][k], it)
            assert same(r["x"][k], x) and same(r["lam"][k], lam) and same(r["fval"][k], fval), (t, k)
        f = f + 0.05 * rng.standard_normal(f.shape)
        mdl.update(f=f)
        for k, om in enumerate(oms):
            om.update(4, f=f[k])
    mdl.close()


def test_golden_proximal_fixtures(gpu_lib, monkeypatch):
    """the reference's own outputs (tests/golden/golden_prox.npz, written by make_golden.py from the strict build)"""
    import os
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_prox.npz"), allow_pickle=False)
    for nm in sorted({k.split("/")[0] for k in g.files} - {"warm"}):
        get = lambda f: g[f"{nm}/{f}"]
        H = get("H") if get("H").size else None    # an LP
        x, fval, flag, info = daqp_amd.solve(H, get("f"), get("A"), get("bupper"), get("blower"), get("sense"),
                                             eps_prox=float(get("eps_prox")), eta_prox=float(get("eta_prox")), iter_limit=int(get("iter_limit")))
        assert flag == int(get("exitflag")), nm
        if flag != -5:
            assert info["iterations"] == int(get("iter")), nm
        if flag > 0:
            assert same(x, get("x")) and same(info["lam"], get("lam")) and fval == float(get("fval")), nm
    n, m, ms = int(g["warm/n"]), int(g["warm/m"]), int(g["warm/ms"])
    mdl = daqp_amd.Model()
    mdl.setup(g["warm/H"], g["warm/fs"][0], g["warm/A"], g["warm/bupper"], g["warm/blower"], np.zeros(m, np.int32))
    for t in range(g["warm/fs"].shape[0]):
        if t > 0:
            mdl.update(f=g["warm/fs"][t])
        x, fval, flag, info = mdl.solve()
        assert flag == int(g["warm/exitflag"][t]) and info["iterations"] == int(g["warm/iter"][t]), t
        assert same(x, g["warm/x"][t]) and same(info["lam"], g["warm/lam"][t]) and fval == float(g["warm/fval"][t]), t


def lp_cases(count, seed, nmax=25):
    out = []
    for k in range(count):
        rng = np.random.default_rng([seed, k])
        n = int(rng.integers(2, nmax)); m = int(rng.integers(n + 1, 3 * n + 3)); ms = int(rng.integers(0, min(n, m) + 1)) if k % 2 else 0
        kw = {}
        if k % 5 == 4:
            kw = dict(eta_prox=1e-9)
        if k % 11 == 5:
            kw = dict(iter_limit=int(rng.integers(2, 15)))
        out.append((O.generate_lp(n, m, ms, [seed + 1, k], unbounded=(k % 7 == 3)), kw))
    return out


def test_linear_programs_single(oracle, gpu_lib, monkeypatch):
    """H = None: the LP branch of daqp_prox.c (R = I, adaptive smoothing, gradient steps, unbounded detection) through
    the drop-in daqp_quadprog, against the oracle (pinned bit for bit on LPs against the reference)."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    flags = set()
    for q, kw in lp_cases(60, 410):
        x, lam, fval, flag, it = oracle.quadprog(None, q["f"], q["A"], q["bupper"], q["blower"], q["sense"], settings=O.default_settings(**kw))
        xg, fg, flg, inf = daqp_amd.solve(None, q["f"], q["A"], q["bupper"], q["blower"], q["sense"], **kw)
        assert flg == flag and inf["iterations"] == it, (kw, flg, flag, inf["iterations"], it)
        flags.add(flag)
        if flag > 0:
            assert same(xg, x) and same(inf["lam"], lam) and fg == fval
    assert flags == {1, -3, -4}


@pytest.mark.parametrize("n,m,ms", [(5, 12, 0), (12, 30, 4), (24, 60, 0), (50, 150, 0), (70, 160, 10), (64, 128, 0), (64, 120, 10)])   # (n = 64: vertices of 64 rows + the exchange = 65, register kernel -> k_ldp hand-over inside the proximal loop)
def test_linear_program_batches(oracle, gpu_lib, monkeypatch, n, m, ms):
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    N = 32
    qs = [O.generate_lp(n, m, ms, [420, n, k], unbounded=(k % 9 == 4)) for k in range(N)]
    ref = [oracle.quadprog(None, q["f"], q["A"], q["bupper"], q["blower"], q["sense"]) for q in qs]
    b = {k: np.stack([q[k] for q in qs]) for k in ("f", "A", "bupper", "blower", "sense")}
    mdl = daqp_amd.BatchModel(N, n, m, ms)
    mdl.setup(None, b["f"], b["A"], b["bupper"], b["blower"], b["sense"], init_mask=64)
    r = mdl.solve()
    info = mdl.prox_info()
    assert (info["n_prox"] == n).all()
    for k in range(N):
        x, lam, fval, flag, it = ref[k]
        assert r["exitflag"][k] == flag and r["iter"][k] == it, (k, r["exitflag"][k], flag, r["iter"][k], it)
        if flag > 0:
            assert same(r["x"][k], x) and same(r["lam"][k], lam) and same(r["fval"][k], fval), k
    # a second solve after new costs: warm working sets, the iterate carries over
    f2 = b["f"] + 0.3 * np.random.default_rng(421).standard_normal(b["f"].shape)
    oms = []
    for k, q in enumerate(qs):
        om = oracle.model(n, m, ms)
        assert om.setup(None, q["f"], q["A"], q["bupper"], q["blower"], q["sense"]) == 1
        om.solve()
        om.update(4, f=f2[k])
        oms.append(om)
    mdl.update(f=f2)
    r2 = mdl.solve()
    for k, om in enumerate(oms):
        x, lam, fval, flag, it = om.solve()[:5]
        assert r2["exitflag"][k] == flag and r2["iter"][k] == it, (k, r2["exitflag"][k], flag, r2["iter"][k], it)
        if flag > 0:
            assert same(r2["x"][k], x) and same(r2["lam"][k], lam) and same(r2["fval"][k], fval), k
    mdl.close()


def test_primal_start(oracle, gpu_lib, monkeypatch):
    """daqp_set_primal_start (api.c:636-641) / daqp_batch_set_primal_start: the proximal iterations from a given point --
    batch API and drop-in workspace, singular QPs and LPs."""
    import ctypes as C
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms, N = 10, 26, 2, 12
    for lp in (False, True):
        qs = [O.generate_lp(n, m, ms, [86, k]) if lp else O.generate_singular_qp(n, m, ms, rank=3 + k % 5, rng=[87, k], in_range=(k % 2 == 0))
              for k in range(N)]
        kw = {} if lp else dict(eps_prox=1e-2, eta_prox=1e-8)
        x0 = np.random.default_rng(88).standard_normal((N, n))
        refs = []
        for k, q in enumerate(qs):
            om = oracle.model(n, m, ms, settings=O.default_settings(**kw))
            assert om.setup(**q) == 1
            om.set_primal_start(x0[k])
            refs.append(om.solve())
        b = {kk: np.stack([q[kk] for q in qs]) for kk in ("f", "A", "bupper", "blower", "sense")}
        H = None if lp else np.stack([q["H"] for q in qs])
        mdl = daqp_amd.BatchModel(N, n, m, ms, **kw)
        mdl.setup(H, b["f"], b["A"], b["bupper"], b["blower"], b["sense"])
        mdl.set_primal_start(x0)
        r = mdl.solve()
        for k in range(N):
            x, lam, fval, flag, it = refs[k][:5]
            assert r["exitflag"][k] == flag and r["iter"][k] == it, (lp, k, r["iter"][k], it)
            if flag > 0:
                assert same(r["x"][k], x) and same(r["lam"][k], lam) and same(r["fval"][k], fval), (lp, k)
        mdl.close()
        # the drop-in workspace
        one = daqp_amd.Model()
        q = qs[0]
        flag, _ = one.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"], **kw)
        assert flag == 1
        xs = np.ascontiguousarray(x0[0])
        daqp_amd.lib().daqp_set_primal_start(one._ws, xs.ctypes.data_as(C.POINTER(C.c_double)))
        x, fval, fl, info = one.solve()
        assert fl == refs[0][3] and info["iterations"] == refs[0][4] and same(x, refs[0][0]) and fval == refs[0][2]


def test_shift_doubling(oracle, gpu_lib, monkeypatch):
    """slightly indefinite Hessians: the first shift is not enough, eps doubles pass after pass (utils.c:357-360) -- a
    different number of passes for different problems of one batch, some never definite (-5)."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms, N = 9, 22, 2, 14
    qs = []
    for k in range(N):
        q = O.generate_singular_qp(n, m, ms, rank=4 + k % 4, rng=[89, k])
        q["H"] = q["H"] - [1e-5, 1e-3, 0.05, 0.7, 3.0, 40.0, 1e5][k % 7] * np.eye(n)
        qs.append(q)
    ref = oracle_each(oracle, qs)
    assert {r[3] for r in ref} >= {1, -5}
    b = stack(qs)
    mdl = daqp_amd.BatchModel(N, n, m, ms)
    mdl.setup(b["H"], b["f"], b["A"], b["bupper"], b["blower"], b["sense"], init_mask=64)
    flags = mdl.setup_flags()
    r = mdl.solve()
    eps = mdl.prox_info()["eps"]
    assert len({float(e) for e in eps if e > 0}) >= 4      # several different numbers of doublings in one batch
    for k in range(N):
        x, lam, fval, flag, it = ref[k]
        assert r["exitflag"][k] == flag, (k, r["exitflag"][k], flag)
        assert (flags[k] == 1) == (flag != -5)
        if flag != -5:
            assert r["iter"][k] == it
        if flag > 0:
            assert same(r["x"][k], x) and same(r["lam"][k], lam) and same(r["fval"][k], fval), k
    mdl.close()


@pytest.mark.parametrize("exact", ["1", "0"])
@pytest.mark.parametrize("shape", [(9, 22, 2), (50, 150, 0), (12, 40, 12), (70, 150, 3)])
def test_failed_shifts_report_their_flag_in_a_one_shot_batch(oracle, gpu_lib, monkeypatch, shape, exact):
    """solve_batch straight after the setup, nothing in between that would resolve the setup flags (the solve launch goes out before the
    host knows of the singular Hessians): problems whose regularising passes fail -- indefinite H: -5 after the doublings; a zero row
    of A whose bounds exclude 0 in a problem that needed the shift: -1 -- report THAT, never the internal "needs the shift" code;
    definite, semidefinite and infeasible problems of the same batch as usual."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", exact)
    n, m, ms = shape
    N = 21
    qs = []
    for k in range(N):
        if k % 3 == 0:
            q = O.generate_qp(n, m, ms, max(1, n // 3), rng=[97, n, k])
            q["sense"] = np.zeros(m, np.int32)
        else:
            q = O.generate_singular_qp(n, m, ms, rank=1 + (k * 5) % (n - 1), rng=[97, n, k])
            if k % 3 == 2:
                q["H"] = q["H"] - [0.7, 40.0, 1e5][(k // 3) % 3] * np.eye(n)        # indefinite: no shift is enough
            elif k % 6 == 1 and m > ms:
                q["A"] = q["A"].copy(); q["A"][0] = 0.0                                  # zero row, 0 outside its bounds
                q["bupper"] = q["bupper"].copy(); q["blower"] = q["blower"].copy()
                q["blower"][ms] = 1.0; q["bupper"][ms] = 2.0
        qs.append(q)
    ref = oracle_each(oracle, qs)
    assert {r[3] for r in ref} >= {1, -5, -1}
    b = stack(qs)
    r = daqp_amd.solve_batch(b["H"], b["f"], b["A"], b["bupper"], b["blower"], b["sense"], ms=ms)
    mdl = daqp_amd.BatchModel(N, n, m, ms)
    mdl.setup(b["H"], b["f"], b["A"], b["bupper"], b["blower"], b["sense"], init_mask=64)
    r2 = mdl.solve()                                             # no setup_flags() / prox_info() before the first solve
    for k in range(N):
        x, lam, fval, flag, it = ref[k]
        for got in (r, r2):
            assert got["exitflag"][k] == flag, (k, got["exitflag"][k], flag)
            if flag > 0 and (exact == "1" or k % 3 != 0):      # (shifted problems are bit-identical in both modes)
                assert got["iter"][k] == it and same(got["x"][k], x) and same(got["lam"][k], lam), k
            elif flag > 0:
                assert got["iter"][k] == it and np.abs(got["x"][k] - x).max() < 1e-9
    assert (mdl.setup_flags() < 0).sum() == sum(1 for rr in ref if rr[3] in (-5,)) + sum(1 for k, rr in enumerate(ref) if rr[3] == -1 and k % 6 == 1)
    mdl.close()


@pytest.mark.parametrize("n,m,ms,kind", [(120, 260, 5, "sing"), (150, 300, 0, "lp"), (200, 420, 10, "diag"), (260, 540, 4, "sing"), (258, 530, 0, "lp")])
def test_large_shapes(oracle, gpu_lib, monkeypatch, n, m, ms, kind):
    """n > 64: the generic setup kernel (factors in LDS or HBM scratch) and the streamed / spilled solve kernels, with the
    spilled instantiation of the LP gradient step."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    N = 4
    if kind == "lp":
        qs = [O.generate_lp(n, m, ms, [90, n, k], unbounded=(k == 3)) for k in range(N)]
    else:
        qs = [O.generate_singular_qp(n, m, ms, rank=n // 2 + 7 * k, rng=[90, n, k], kind="diag" if kind == "diag" else "dense") for k in range(N)]
    ref = [oracle.quadprog(q.get("H"), q["f"], q["A"], q["bupper"], q["blower"], q["sense"]) for q in qs]
    b = {k: np.stack([q[k] for q in qs]) for k in ("f", "A", "bupper", "blower", "sense")}
    H = None if kind == "lp" else np.stack([q["H"] for q in qs])
    r = daqp_amd.solve_batch(H, b["f"], b["A"], b["bupper"], b["blower"], b["sense"], ms=ms)
    for k in range(N):
        x, lam, fval, flag, it = ref[k]
        assert r["exitflag"][k] == flag and r["iter"][k] == it, (k, r["exitflag"][k], flag, r["iter"][k], it)
        if flag > 0:
            assert same(r["x"][k], x) and same(r["lam"][k], lam) and same(r["fval"][k], fval), k


def test_only_simple_bounds(oracle, gpu_lib, monkeypatch):
    """m == ms: no general rows at all (A is empty) -- a box LP ends at the vertex picked by the signs of f, a singular QP
    moves only inside its box."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n = 7
    rng = np.random.default_rng(91)
    for lp in (True, False):
        N = 6
        f = rng.standard_normal((N, n))
        bu, bl = 0.5 + rng.random((N, n)), -0.5 - rng.random((N, n))
        A = np.zeros((N, 0, n))
        H = None
        if not lp:
            T = rng.standard_normal((N, 3, n))
            H = np.einsum("qri,qrj->qij", T, T)
        ref = [oracle.quadprog(None if lp else H[k], f[k], np.zeros((0, n)), bu[k], bl[k], np.zeros(n, np.int32)) for k in range(N)]
        r = daqp_amd.solve_batch(H, f, A, bu, bl, None, ms=n)
        for k in range(N):
            x, lam, fval, flag, it = ref[k]
            assert r["exitflag"][k] == flag == 1 and r["iter"][k] == it, (lp, k, r["exitflag"][k], flag, r["iter"][k], it)
            assert same(r["x"][k], x) and same(r["lam"][k], lam) and same(r["fval"][k], fval), (lp, k)
            if lp:
                assert np.abs(x - np.where(f[k] > 0, bl[k], bu[k])).max() < 1e-9


def test_concurrent_host_threads_proximal(oracle, gpu_lib, monkeypatch):
    """the outer loop is host-driven (launches and read-backs per outer iteration): four host threads, each with its own
    batch of singular QPs / LPs, must get the answers of a serial run"""
    import threading
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms, N = 11, 28, 2, 24
    probs = []
    for t in range(4):
        if t % 2:
            qs = [O.generate_lp(n, m, ms, [92, t, k]) for k in range(N)]
        else:
            qs = [O.generate_singular_qp(n, m, ms, rank=3 + k % 6, rng=[93, t, k], in_range=(k % 2 == 0)) for k in range(N)]
        probs.append(qs)
    out = [None] * 4

    def work(t):
        qs = probs[t]
        b = {k: np.stack([q[k] for q in qs]) for k in ("f", "A", "bupper", "blower", "sense")}
        H = None if t % 2 else np.stack([q["H"] for q in qs])
        for _ in range(3):
            out[t] = daqp_amd.solve_batch(H, b["f"], b["A"], b["bupper"], b["blower"], b["sense"], ms=ms)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for t in range(4):
        for k, q in enumerate(probs[t]):
            x, lam, fval, flag, it = oracle.quadprog(q.get("H"), q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
            assert out[t]["exitflag"][k] == flag and out[t]["iter"][k] == it, (t, k)
            if flag > 0:
                assert same(out[t]["x"][k], x) and same(out[t]["lam"][k], lam), (t, k)


@pytest.mark.parametrize("exact", ["1", "0"])
@pytest.mark.parametrize("n,m,ms,kind", [(12, 30, 4, "dense"), (20, 60, 0, "diag"), (50, 150, 0, "dense"), (70, 150, 6, "dense")])
def test_shared_singular_hessian(oracle, gpu_lib, monkeypatch, n, m, ms, kind, exact):
    """ONE positive semi-definite H and ONE A for the whole batch (daqp_batch_setup_shared): the regularising passes run on the
    one factorisation (utils.c:354-377), then every problem iterates the proximal loop (daqp_prox.c:21-221) on the shared shifted
    factor with its own centre -- the reference's MPC usage (setup_daqp with open bounds, daqp_update_ldp(UPDATE_v|UPDATE_d),
    daqp_solve) per problem, bit for bit, through a warm re-solve as well."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", exact)
    N = 24
    q0 = O.generate_singular_qp(n, m, ms, rank=max(1, n // 2), rng=[91, n], kind=kind)
    rng = np.random.default_rng([92, n])
    f = q0["f"][None, :] + 0.2 * rng.standard_normal((N, n))
    shift = 0.03 * rng.standard_normal((N, m))
    bu, bl = q0["bupper"][None, :] + shift, q0["blower"][None, :] + shift
    bm = daqp_amd.BatchModel(N, n, m, ms)
    # (the second shape passes an all-zero sense array: the eager route -- update kernel and activation pass at setup)
    bm.setup_shared(q0["H"], f, q0["A"], bu, bl, np.zeros((N, m), np.int32) if n == 20 else None)
    info = bm.prox_info()
    assert (info["n_prox"] > 0).all() and (info["eps"] > 0).all() and len(set(info["eps"].tolist())) == 1
    models = []
    for k in range(N):
        om = oracle.model(n, m, ms)
        assert om.setup(q0["H"], f[k], q0["A"], np.full(m, 1e30), np.full(m, -1e30), None) >= 0
        assert om.update(O.UPDATE_v | O.UPDATE_d, f=f[k], bupper=bu[k], blower=bl[k]) == 0
        models.append(om)
    good, outer = 0, 0
    for t in range(2):
        if t > 0:
            f = f + 0.05 * rng.standard_normal((N, n))
            bm.update(f=f)
            for k in range(N):
                assert models[k].update(O.UPDATE_v, f=f[k]) == 0
        g = bm.solve()
        outer = max(outer, int(bm.prox_info()["outer"].max()))
        for k in range(N):
            r = models[k].solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (t, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
            if r[3] > 0:
                good += 1
                assert same(g["x"][k], r[0]) and same(g["lam"][k], r[1]) and same(g["fval"][k], r[2]), (t, k)
    assert good > N and outer >= 2, (good, outer)
    bm.close()
