"""The reference's own known-answer tests, run on the HIP path through the drop-in C ABI
(daqp_quadprog / setup_daqp / daqp_solve / daqp_update_ldp), each cross-checked against the oracle:
  * hand examples of the Python binding's tests            interfaces/daqp-python/test/example_test.py:17-26,175-237
  * generator QPs with analytic optimum, KKT and fval      interfaces/daqp-julia/test/core_tests.jl:26-30, core_test.m:16-26
  * iter_limit = 1 -> -4                                    core_tests.jl:33-35
  * exact warm start -> exactly one iteration               core_tests.jl:520-545
  * unconstrained optimum inside the bounds (shortcut)      core_tests.jl:825-839
  * crossed bounds -> -1                                    core_test.m:212-221
plus the edge cases of the batch boundary (empty batch, one QP, no general rows, only equalities, soft rows)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def exact_mode(monkeypatch):
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float64).view(np.uint64), np.ascontiguousarray(b, np.float64).view(np.uint64))


def test_hand_examples(oracle, gpu_lib):
    import daqp_amd
    H, f, A = np.eye(2), np.array([2.0, 2.0]), np.zeros((0, 2))
    for ff, bu, want in ((f, 1.0, [-1, -1]), (-f, 1.0, [1, 1]), (f, 0.5, [-0.5, -0.5])):
        x, fval, flag, info = daqp_amd.solve(H, ff, A, bu * np.ones(2), -bu * np.ones(2), np.zeros(2, np.int32))
        r = oracle.quadprog(H, ff, A, bu * np.ones(2), -bu * np.ones(2), np.zeros(2, np.int32))
        assert flag == 1 and np.allclose(x, want, atol=1e-6) and same(x, r[0]) and info["iterations"] == r[4]
    # example_test.py:17-26: H = [[1,0],[0,1]], f = [1,1], A = [[1,2],[1,-1]], bupper = [1,2,3,4], blower = [-1,-2,-3,-4]
    H = np.eye(2); f = np.ones(2); A = np.array([[1.0, 2.0], [1.0, -1.0]])
    bu = np.array([1.0, 2, 3, 4]); bl = -bu
    x, fval, flag, info = daqp_amd.solve(H, f, A, bu, bl, np.zeros(4, np.int32))
    r = oracle.quadprog(H, f, A, bu, bl, np.zeros(4, np.int32))
    assert flag == r[3] == 1 and same(x, r[0]) and fval == r[2] and np.allclose(x, [-1, -1], atol=1e-9)


def test_generator_optimum_kkt_fval(oracle, gpu_lib):
    import daqp_amd
    for cfg in ("C1", "C3"):
        n, m, ms, na, seed, _ = O.CONFIGS[cfg]
        for k in range(10):
            q = O.generate_qp(n, m, ms, na, rng=[seed, 100 + k])
            x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
            lam = info["lam"]
            assert flag == 1 and np.abs(x - q["x"]).max() < 1e-8 and np.count_nonzero(lam) == na
            Afull = np.vstack([np.eye(n)[:ms], q["A"]])
            assert np.abs(q["H"] @ x + q["f"] + Afull.T @ lam).max() < 1e-7
            assert abs(0.5 * x @ q["H"] @ x + q["f"] @ x - fval) < 1e-8
            Ax = Afull @ x
            assert (Ax <= q["bupper"] + 1e-6).all() and (Ax >= q["blower"] - 1e-6).all()
            assert (lam[Ax < q["bupper"] - 1e-6] <= 0).all() and (lam[Ax > q["blower"] + 1e-6] >= 0).all()


def test_iteration_limit_and_crossed_bounds(oracle, gpu_lib):
    import daqp_amd
    q = O.generate_qp(20, 40, 0, 8, rng=[1234, 0])
    x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, iter_limit=1)
    assert flag == -4
    bu = q["bupper"].copy(); bu[3] = q["blower"][3] - 1
    x0 = np.full(20, 7.0)
    x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], bu, q["blower"], None)
    assert flag == -1 == oracle.quadprog(q["H"], q["f"], q["A"], bu, q["blower"])[3]


def test_exact_warm_start_needs_one_iteration(oracle, gpu_lib):
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS["C1"]
    q = O.generate_qp(n, m, ms, na, rng=[seed, 3])
    x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    lam = info["lam"]
    sense = np.zeros(m, np.int32)
    sense[lam > 1e-12] |= O.ACTIVE
    sense[lam < -1e-12] |= O.ACTIVE + O.LOWER
    x2, fval2, flag2, info2 = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], sense)
    r = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], sense)
    assert flag2 == 1 and info2["iterations"] == 1 == r[4] and same(x2, r[0]) and np.abs(x - x2).max() < 1e-10


def test_unconstrained_shortcut(oracle, gpu_lib):
    """bounds far from the unconstrained optimum: solved in setup (daqp_check_unconstrained), iter = 1, lam = 0"""
    import daqp_amd
    q = O.generate_qp(12, 30, 4, 5, rng=[7, 7])
    bu, bl = np.full(30, 1e3), np.full(30, -1e3)
    x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], bu, bl, None)
    r = oracle.quadprog(q["H"], q["f"], q["A"], bu, bl, None)
    assert flag == r[3] == 1 and info["iterations"] == r[4] == 1 and not info["lam"].any() and same(x, r[0])
    assert np.abs(x + np.linalg.solve(q["H"], q["f"])).max() < 1e-9
    H = np.eye(3)
    x, fval, flag, info = daqp_amd.solve(H, np.zeros(3), np.zeros((0, 3)), np.ones(3), -np.ones(3), None)
    assert flag == 1 and not x.any()      # core_tests.jl:825-839: x = 0


def test_boundary_edge_cases(oracle, gpu_lib):
    import daqp_amd
    # empty batch
    g = daqp_amd.solve_batch(np.zeros((0, 4, 4)), np.zeros((0, 4)), np.zeros((0, 3, 4)), np.zeros((0, 5)), np.zeros((0, 5)), None, ms=2)
    assert g["x"].shape == (0, 4) and g["exitflag"].shape == (0,)
    # a batch of one, only simple bounds (no general rows at all)
    q = O.generate_qp(6, 6, 6, 3, rng=[5, 1])
    g = daqp_amd.solve_batch(q["H"][None], q["f"][None], np.zeros((1, 0, 6)), q["bupper"][None], q["blower"][None], None, ms=6)
    r = oracle.quadprog(q["H"], q["f"], np.zeros((0, 6)), q["bupper"], q["blower"], None)
    assert g["exitflag"][0] == r[3] and g["iter"][0] == r[4] and same(g["x"][0], r[0]) and same(g["lam"][0], r[1])
    # equalities (sense 5) and soft rows (sense 8): exit flag 1 / 2 as the reference decides
    for trial in range(20):
        rng = np.random.default_rng([31, trial])
        q = O.generate_nasty(8, 20, 2, 3, 1e-3, rng, n_dup=0, n_eq=2, n_soft=2)
        x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        r = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        assert flag == r[3] and info["iterations"] == r[4]
        if flag > 0:
            assert same(x, r[0]) and same(info["lam"], r[1])
    # infeasible: two parallel rows that exclude each other
    H = np.eye(2); A = np.array([[1.0, 0.0], [1.0, 0.0]])
    x, fval, flag, info = daqp_amd.solve(H, np.zeros(2), A, np.array([1.0, -2.0]), np.array([0.5, -3.0]), np.zeros(2, np.int32))
    assert flag == -1 == oracle.quadprog(H, np.zeros(2), A, np.array([1.0, -2.0]), np.array([0.5, -3.0]), np.zeros(2, np.int32))[3]
    # non-convex Hessian -> -5 with eps_prox = 0 (api.c / utils.c:356-377)
    Hn = np.array([[1.0, 2.0], [2.0, 1.0]])
    x, fval, flag, info = daqp_amd.solve(Hn, np.zeros(2), np.zeros((0, 2)), np.ones(2), -np.ones(2), None, eps_prox=0.0)
    assert flag == -5


def test_concurrent_host_threads(oracle, gpu_lib):
    """the reference is re-entrant per workspace (SURVEY 8b "Threading"): four host threads, each with its own batch
    (own HIP stream), must get the answers of a serial run"""
    import threading
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS["C1"]
    qs = [O.generate_batch(96, n, m, ms, na, seed, start=1000 * t) for t in range(4)]
    out = [None] * 4

    def work(t):
        q = qs[t]
        for _ in range(3):
            out[t] = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for t in range(4):
        q = qs[t]
        r = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
        assert np.array_equal(out[t]["exitflag"], r[3]) and np.array_equal(out[t]["iter"], r[4])
        assert same(out[t]["x"], r[0]) and same(out[t]["lam"], r[1])


def test_concurrent_host_threads_single_problem_symbols(oracle, gpu_lib):
    """the drop-in symbols themselves from four host threads at once (SURVEY 8b "Threading"; the binding releases the GIL around
    them, daqp.pyx:211-212,467-468): one-shot daqp_quadprog calls of ONE shape -- so the threads compete for the parked one-problem
    workspaces, their mapped result slabs and completion words -- interleaved with a kept workspace per thread that is updated
    (UPDATE_v, the deferred update of the latency path) and re-solved; every answer equal to the serial oracle's"""
    import threading
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS["C1"]
    T, K = 4, 12
    qs = [[O.generate_qp(n, m, ms, na, rng=[seed, 7000 + 100 * t + k]) for k in range(K)] for t in range(T)]
    got = [[None] * K for _ in range(T)]
    warm = [[None] * K for _ in range(T)]
    err = []

    def work(t):
        try:
            mdl = daqp_amd.Model()
            q0 = qs[t][0]
            mdl.setup(q0["H"], q0["f"], q0["A"], q0["bupper"], q0["blower"], np.zeros(m, np.int32))
            mdl.solve()
            for k in range(K):
                q = qs[t][k]
                x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], np.zeros(m, np.int32))
                got[t][k] = (x, info["lam"], flag, info["iterations"])
                mdl.update(f=q["f"])
                x, fval, flag, info = mdl.solve()
                warm[t][k] = (x, info["lam"], flag, info["iterations"])
        except Exception as e:   # noqa: BLE001 (reported below, on the main thread)
            err.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for t in range(T):
        om = O.OracleModel(oracle, n, m, ms)
        q0 = qs[t][0]
        om.setup(q0["H"], q0["f"], q0["A"], q0["bupper"], q0["blower"], np.zeros(m, np.int32))
        om.solve()
        for k in range(K):
            q = qs[t][k]
            r = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], np.zeros(m, np.int32))
            x, lam, flag, it = got[t][k]
            assert flag == r[3] and it == r[4] and same(x, r[0]) and same(lam, r[1]), (t, k)
            om.update(O.UPDATE_v, f=q["f"])
            w = om.solve()
            x, lam, flag, it = warm[t][k]
            assert flag == w[3] and it == w[4] and same(x, w[0]) and same(lam, w[1]), (t, k, "warm")


@pytest.mark.parametrize("shape", [(50, 150, 0, 20), (12, 48, 12, 6), (20, 40, 0, 8), (80, 200, 5, 30)])   # (the last one: generic setup + workgroup kernel)
def test_shared_structure_batch(oracle, gpu_lib, shape):
    """condensed-MPC batches: ONE H and A, per-problem f and bounds.  daqp_batch_setup_shared is the
    reference's own MPC usage batched: factor once (setup_daqp with open bounds), then per problem
    daqp_update_ldp(UPDATE_v|UPDATE_d) with its f / bounds and solve -- bit for bit, including later warm updates."""
    import daqp_amd
    n, m, ms, na = shape
    N = 40
    q0 = O.generate_qp(n, m, ms, na, rng=[811, n])
    rng = np.random.default_rng([812, n])
    f = q0["f"][None, :] + 0.3 * rng.standard_normal((N, n))
    shift = 0.05 * rng.standard_normal((N, m))
    bu, bl = q0["bupper"][None, :] + shift, q0["blower"][None, :] + shift
    bm = daqp_amd.BatchModel(N, n, m, ms)
    # the third shape passes an (all-zero) sense array: that takes the eager route (update kernel + activation pass at setup)
    bm.setup_shared(q0["H"], f, q0["A"], bu, bl, np.zeros((N, m), np.int32) if n == 20 else None)
    models = []
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(q0["H"], f[k], q0["A"], np.full(m, 1e30), np.full(m, -1e30), None)
        assert om.update(O.UPDATE_v | O.UPDATE_d, f=f[k], bupper=bu[k], blower=bl[k]) == 0
        models.append(om)
    for t in range(3):
        if t > 0:
            f = f + 0.05 * rng.standard_normal((N, n))
            if t == 2:
                sh = 0.02 * rng.standard_normal((N, m)); bu, bl = bu + sh, bl + sh
                bm.update(f=f, bupper=bu, blower=bl)
            else:
                bm.update(f=f)
            for k in range(N):
                if t == 2:
                    assert models[k].update(O.UPDATE_v | O.UPDATE_d, f=f[k], bupper=bu[k], blower=bl[k]) == 0
                else:
                    assert models[k].update(O.UPDATE_v, f=f[k]) == 0
        g = bm.solve()
        for k in range(N):
            r = models[k].solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (t, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
            if r[3] > 0:
                assert same(g["x"][k], r[0]) and same(g["lam"][k], r[1])
    bm.close()


@pytest.mark.parametrize("shape", [(10, 30, 4, 4), (16, 40, 0, 6), (33, 80, 10, 10), (50, 150, 0, 20), (70, 150, 6, 14)])
def test_diagonal_hessian_bitwise(oracle, gpu_lib, shape):
    """a diagonal H takes the reference's RinvD branch (utils.c:245-312,455-468,479-480,527-531; daqp.c:130-134;
    auxiliary.c:57-64,104-106): RinvD_i = 1/sqrt(H_ii), unit rows for the simple bounds, scaling_i = sqrt(H_ii).  Bit
    for bit through daqp_quadprog semantics (batch), the persistent-workspace path and a warm update; identity H too."""
    import daqp_amd
    n, m, ms, na = shape
    N = 32
    q = O.generate_batch(N, n, m, ms, na, 5000 + n)
    rng = np.random.default_rng([77, n])
    H = np.zeros((N, n, n))
    for k in range(N):
        H[k] = np.diag(np.ones(n) if k % 4 == 0 else rng.uniform(0.3, 4.0, n))
    r = oracle.quadprog_batch(H, q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(H, q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(g["exitflag"], r[3]) and np.array_equal(g["iter"], r[4])
    ok = r[3] > 0
    assert ok.any() and same(g["x"][ok], r[0][ok]) and same(g["lam"][ok], r[1][ok]) and same(g["fval"][ok], r[2][ok])
    # persistent workspaces: setup (no unconstrained shortcut), solve, update f, solve
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.setup(H, q["f"], q["A"], q["bupper"], q["blower"])
    models = []
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(H[k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        models.append(om)
    f = q["f"].copy()
    for t in range(2):
        if t:
            f = f + 0.05 * rng.standard_normal(f.shape)
            bm.update(f=f)
            for k in range(N):
                assert models[k].update(O.UPDATE_v, f=f[k]) == 0
        gg = bm.solve()
        for k in range(N):
            rr = models[k].solve()
            assert gg["exitflag"][k] == rr[3] and gg["iter"][k] == rr[4], (t, k)
            if rr[3] > 0:
                assert same(gg["x"][k], rr[0]) and same(gg["lam"][k], rr[1])
    bm.close()


def test_shared_structure_with_sense(oracle, gpu_lib):
    """shared-structure batch whose problems carry equality (5) and soft (8) rows: the working set given by sense is
    activated at setup (eager update + activation pass); results as the oracle's setup(open bounds, sense) + update + solve"""
    import daqp_amd
    N = 24
    for trial in range(6):
        rng = np.random.default_rng([91, trial])
        q0 = O.generate_nasty(10, 26, 3, 4, 1e-2, rng, n_dup=0, n_eq=2, n_soft=2)
        n, m, ms = 10, 26, 3
        f = q0["f"][None, :] + 0.1 * rng.standard_normal((N, n))
        bu = np.repeat(q0["bupper"][None, :], N, 0); bl = np.repeat(q0["blower"][None, :], N, 0)
        sense = np.repeat(q0["sense"][None, :], N, 0).astype(np.int32)
        bm = daqp_amd.BatchModel(N, n, m, ms, ns_max=2)
        bm.setup_shared(q0["H"], f, q0["A"], bu, bl, sense)
        g = bm.solve()
        for k in range(N):
            om = oracle.model(n, m, ms)
            om.setup(q0["H"], f[k], q0["A"], np.full(m, 1e30), np.full(m, -1e30), q0["sense"])
            assert om.update(O.UPDATE_v | O.UPDATE_d, f=f[k], bupper=bu[k], blower=bl[k]) == 0
            r = om.solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (trial, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
            if r[3] > 0:
                assert same(g["x"][k], r[0]) and same(g["lam"][k], r[1])
        bm.close()


def test_equality_heavy_quadprog_against_reference_fixtures(gpu_lib):
    """daqp_quadprog on QPs with many equalities: the reference eliminates them before solving (eq_elim.c:127-164, more than
    5 equalities and 10 neq > n); this library solves the full LDP.  Same exit flag and active set, x and lam to the
    north-star tolerance; the iteration count is the one documented difference.  Fixtures: outputs of the reference
    (tests/golden/golden_eliminated.npz, make_golden.py)."""
    import os
    import daqp_amd
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_eliminated.npz"), allow_pickle=False)
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) == 16
    for nm in names:
        get = lambda f: g[f"{nm}/{f}"]
        x, fval, flag, info = daqp_amd.solve(get("H"), get("f"), get("A"), get("bupper"), get("blower"), get("sense"))
        assert flag == int(get("exitflag")) == 1, nm
        assert np.abs(x - get("x")).max() < 1e-9, nm
        assert np.abs(info["lam"] - get("lam")).max() < 1e-7 * max(1.0, np.abs(get("lam")).max()), nm
        ineq = get("sense") != 5
        assert np.array_equal(np.sign(info["lam"][ineq]), np.sign(get("lam")[ineq])), nm
        assert abs(fval - float(get("fval"))) < 1e-9 * max(1.0, abs(float(get("fval")))), nm
    # batched, default arithmetic
    sel = [nm for nm in names if g[f"{nm}/H"].shape[0] == g[f"{names[0]}/H"].shape[0] and g[f"{nm}/bupper"].size == g[f"{names[0]}/bupper"].size]
    nm = sel[0]
    q = {k: np.stack([g[f"{nm}/{k}"]] * 3) for k in ("H", "f", "A", "bupper", "blower", "sense")}
    r = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    assert (r["exitflag"] == 1).all() and np.abs(r["x"] - g[f"{nm}/x"]).max() < 1e-9
