"""HIP path vs the oracle on the same seeded inputs (run with -m gpu on an MI355X).

Bar (BASELINE.json north_star): active sets identical (index AND side, i.e. sign(lam)), identical
iteration counts and exit flags, |x - x_ref|_inf < 1e-9.  Because the kernels keep the reference's
operation order for every sum, the tests additionally demand bit-identical x / lam / fval.
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
XTOL = 1e-9  # north_star tolerance on x*


@pytest.fixture(autouse=True)
def exact_mode(monkeypatch):
    """This file asserts BIT-identical results, which needs the reference's summation order in
    M = A R^-1 (DAQP_AMD_EXACT=1: VALU path).  The default MFMA path is covered by
    tests/test_gpu_fast_mode.py at the north_star bar (identical active sets, |dx| < 1e-9)."""
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float64).view(np.uint64),
                          np.ascontiguousarray(b, np.float64).view(np.uint64))


def gpu_batch(q, **settings):
    import daqp_amd
    return daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q.get("sense"), ms=q["ms"], **settings)


def check_batch(oracle, cfg, N, start=0, bitwise=True, **settings):
    n, m, ms, na, seed, _ = O.CONFIGS[cfg] if isinstance(cfg, str) else cfg
    q = O.generate_batch(N, n, m, ms, na, seed, start=start)
    q["ms"] = ms
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms,
                                settings=O.default_settings(**settings) if settings else None)
    g = gpu_batch(q, **settings)
    assert np.array_equal(g["exitflag"], ref[3]), f"exit flags differ: {np.nonzero(g['exitflag'] != ref[3])[0][:10]}"
    assert np.array_equal(g["iter"], ref[4]), f"iteration counts differ at {np.nonzero(g['iter'] != ref[4])[0][:10]}"
    assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1])), "active sets differ"
    assert np.abs(g["x"] - ref[0]).max() < XTOL
    ok = g["exitflag"] == 1
    assert np.abs(g["x"][ok] - q["xref"][ok]).max(initial=0) < 1e-6   # generator's analytic optimum (core_tests.jl:26-30 uses 1e-4)
    if bitwise:
        assert bits_equal(g["x"], ref[0]) and bits_equal(g["lam"], ref[1]) and bits_equal(g["fval"], ref[2])
    return g, ref


def test_c1_single_qp(oracle, gpu_lib):
    """config 1 shape (n=20, m=40) through the single-problem drop-in entry point daqp_quadprog"""
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS["C1"]
    q = O.generate_qp(n, m, ms, na, rng=[seed, 0])
    x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    r = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    assert flag == r[3] == 1 and info["iterations"] == r[4]
    assert bits_equal(x, r[0]) and bits_equal(info["lam"], r[1]) and fval == r[2]
    assert np.abs(x - q["x"]).max() < 1e-8


@pytest.mark.parametrize("cfg,N", [("C1", 64), ("C2", 256), ("C3", 512)])
def test_config_batches(oracle, gpu_lib, cfg, N):
    check_batch(oracle, cfg, N)


@pytest.mark.parametrize("shape", [(7, 20, 3, 3), (16, 40, 4, 6), (31, 64, 0, 10), (32, 64, 0, 12), (33, 70, 5, 12),
                                   (48, 100, 0, 16), (56, 120, 0, 20), (57, 120, 0, 20), (64, 128, 0, 24)])
def test_setup_shapes(oracle, gpu_lib, shape):
    """every variant of the register setup kernel (NMAX 16/32/56/64), both tile paths: direct HBM->LDS copy (n even,
    not a multiple of 32) and register staging (odd n, n = 32, 64), partial column groups, simple bounds"""
    n, m, ms, na = shape
    check_batch(oracle, (n, m, ms, na, 900 + n, 0), 48)


@pytest.mark.parametrize("shape", [(17, 64, 0, 8), (21, 33, 4, 7), (25, 64, 3, 9), (26, 60, 0, 10), (26, 64, 26, 10)])
def test_three_wave_register_shape(oracle, gpu_lib, shape):
    """k_ldp_reg<1, 13, *>: n = 17 ... 26 with at most one row block, three waves per SIMD (a few registers in scratch) -- the shape class
    of config C1; both arithmetic modes (exact: bit for bit), simple bounds up to ms = n, odd n"""
    n, m, ms, na = shape
    check_batch(oracle, (n, m, ms, na, 1300 + n + ms, 0), 96)


def test_c4_workgroup_kernel(oracle, gpu_lib):
    """config C4 (n=200, m=600): the workgroup-per-problem solve kernel (wg_kernel.hip.h: packed L in LDS, scan / primal
    step / Gram column spread over the waves) -- 64 QPs, bit for bit in exact mode"""
    check_batch(oracle, "C4", 64)


def test_c4_one_wave_kernel(oracle, gpu_lib, monkeypatch):
    """the one-wave generic kernel with L and the active-row cache in HBM scratch (what a problem falls back to when its
    working set outgrows the workgroup kernel's LDS)"""
    monkeypatch.setenv("DAQP_AMD_NO_WG", "1")
    check_batch(oracle, "C4", 6)


@pytest.mark.parametrize("capl", [60, 110])
def test_workgroup_kernel_hands_over_large_working_sets(oracle, gpu_lib, monkeypatch, capl):
    """packed L capped at `capl` rows in LDS: C4 working sets peak at 130-160 rows, so most (60) or some (110) problems are
    flagged during the solve and redone by the one-wave kernel from their untouched state -- results unchanged, bit for bit"""
    monkeypatch.setenv("DAQP_AMD_WG_CAPL", str(capl))
    check_batch(oracle, "C4", 12)


@pytest.mark.parametrize("shape", [(65, 150, 0, 30), (100, 260, 7, 40), (128, 300, 0, 50), (129, 200, 10, 30), (229, 400, 20, 60), (110, 330, 0, 35), (114, 400, 3, 45)])
def test_workgroup_kernel_shapes(oracle, gpu_lib, shape):
    """working sets of 66 ... 230 rows (two- and four-chunk masters), simple bounds, odd n, fewer row blocks than waves; the last two: more row
    blocks than the four waves of a workgroup that shares its CU with a second one (factor within half the LDS)"""
    n, m, ms, na = shape
    check_batch(oracle, (n, m, ms, na, 1900 + n, 0), 10)


def test_workgroup_kernel_event_trace_and_warm_sequence(oracle, gpu_lib):
    """n=100: the add/remove sequence of the reference step for step, then warm updates of f and of the bounds
    (k_update + the workgroup kernel from the stored factors and working set)"""
    import daqp_amd
    n, m, ms, na = 100, 260, 7, 40
    N, T = 6, 3
    q = O.generate_batch(N, n, m, ms, na, 2100)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.enable_trace(4096)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
    models = []
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        om.enable_trace()
        models.append(om)
    f, bu, bl = q["f"].copy(), q["bupper"].copy(), q["blower"].copy()
    for t in range(T + 1):
        if t > 0:
            for k in range(N):
                rng = np.random.default_rng([49, k, t])
                f[k] = f[k] + 0.05 * rng.standard_normal(n)
                shift = 0.02 * rng.standard_normal(m)
                bu[k] = bu[k] + shift; bl[k] = bl[k] + shift
                assert models[k].update(O.UPDATE_v | O.UPDATE_d, f=f[k], bupper=bu[k], blower=bl[k]) == 0
                models[k].enable_trace()
            bm.update(f=f, bupper=bu, blower=bl)
        g = bm.solve()
        tr = bm.read_trace(marks=True)
        for k in range(N):
            r = models[k].solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (t, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
            assert np.array_equal(tr[k], models[k].get_trace(marks=True)), (t, k)
            assert bits_equal(g["x"][k], r[0]) and bits_equal(g["lam"][k], r[1]) and g["fval"][k] == r[2]
    bm.close()


def test_iteration_limit(oracle, gpu_lib):
    g, ref = check_batch(oracle, "C1", 8, bitwise=False, iter_limit=5)
    assert (g["exitflag"] == -4).any()


def test_ldp_setup_bitwise(oracle, gpu_lib):
    """QP -> LDP transform (Cholesky, R^-1, M, v, d, scaling) against the oracle's, bit for bit"""
    import daqp_amd
    for cfg in ("C2", "C3"):
        n, m, ms, na, seed, _ = O.CONFIGS[cfg]
        N = 4
        q = O.generate_batch(N, n, m, ms, na, seed)
        bm = daqp_amd.BatchModel(N, n, m, ms)
        bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
        assert (bm.setup_flags() == 1).all()
        for k in range(N):
            om = oracle.model(n, m, ms)
            assert om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None) == 1
            for name, a, b in zip(("M", "Rinv", "v", "dupper", "dlower", "scaling"), bm.read_ldp(k), om.ldp()):
                assert bits_equal(a, b), f"{cfg}[{k}] {name}: max diff {np.abs(a - b).max():.3e}"
        bm.close()


def test_event_trace(oracle, gpu_lib):
    """the sequence of constraint additions/removals is the reference's, step for step"""
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS["C2"]
    N = 8
    q = O.generate_batch(N, n, m, ms, na, seed, start=1000)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.enable_trace(2048)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
    bm.solve()
    traces = bm.read_trace()
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.enable_trace()
        om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        om.solve()
        assert np.array_equal(traces[k], om.get_trace())
    bm.close()


def test_degenerate_cases(oracle, gpu_lib):
    """near-duplicate rows, dependent equalities, soft rows: pivoting / singular / refine / repair paths"""
    import daqp_amd
    mism = []
    for trial in range(400):
        rng = np.random.default_rng([99, trial])
        eps = 10.0 ** rng.uniform(-13, -2)
        n = int(rng.integers(4, 16)); m = int(rng.integers(n + 4, 4 * n)); ms = int(rng.integers(0, min(n, m // 3) + 1))
        na = int(rng.integers(1, min(n, m - ms)))
        q = O.generate_nasty(n, m, ms, na, eps, rng, n_dup=int(rng.integers(0, 5)), n_eq=int(rng.integers(0, 3)),
                             n_soft=int(rng.integers(0, 3)), dep_eq=bool(rng.integers(0, 2)))
        x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        r = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        ok = flag == r[3] and info["iterations"] == r[4]
        if ok and flag > 0:
            ok = bits_equal(x, r[0]) and bits_equal(info["lam"], r[1])
        if not ok:
            mism.append((trial, flag, r[3], info["iterations"], r[4]))
    assert not mism, mism[:10]


@pytest.mark.parametrize("exact", ["1", "0"])
def test_degenerate_cases_workgroup_kernel(oracle, gpu_lib, monkeypatch, exact):
    """the same families at n = 65 ... 130 (working sets beyond 64 rows: the workgroup kernel with two- and four-chunk
    masters): near-duplicate rows at relative distance 1e-13 ... 1e-2, equalities (some dependent), soft rows.  Exact mode: bit
    for bit; default mode: exit flag and iterations identical, x to 1e-9, the multipliers' combined effect A'lam to 1e-7.  The trace markers say which branches ran."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", exact)
    mism, marks = [], set()
    for trial in range(36):
        rng = np.random.default_rng([199, trial])
        eps = 10.0 ** rng.uniform(-13, -2)
        n = int(rng.integers(65, 131)); m = int(rng.integers(n + 20, 3 * n)); ms = int(rng.integers(0, n // 3))
        na = int(rng.integers(n // 4, n - 4))
        q = O.generate_nasty(n, m, ms, na, eps, rng, n_dup=int(rng.integers(0, 6)), n_eq=int(rng.integers(0, 4)),
                             n_soft=int(rng.integers(0, 4)), dep_eq=bool(rng.integers(0, 2)))
        ns = int((q["sense"] & 8).astype(bool).sum())
        bm = daqp_amd.BatchModel(1, n, m, ms, ns_max=ns)
        bm.enable_trace(1 << 15)
        bm.setup(q["H"][None], q["f"][None], q["A"][None], q["bupper"][None], q["blower"][None], q["sense"][None],
                 init_mask=daqp_amd.UPDATE_unconstrained)
        g = bm.solve()
        tr = bm.read_trace(marks=True)[0]
        marks |= {int(e) for e in tr if abs(int(e)) >= daqp_amd.api.TRACE_MARK}
        bm.close()
        r = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        flag, it = int(g["exitflag"][0]), int(g["iter"][0])
        ok = flag == r[3] and it == r[4]
        if ok and flag > 0:
            if exact == "1":
                ok = bits_equal(g["x"][0], r[0]) and bits_equal(g["lam"][0], r[1])
            else:
                # (near-duplicate active rows at relative distance down to 1e-13 share their multiplier in an ill-determined
                #  way: lam may differ between two arithmetics by O(1) on such a pair while x agrees to 1e-14 -- compare x and
                #  the multipliers' effect  sum_i lam_i [I; A]_i  =  -(H x + f))
                G = np.vstack([np.eye(n)[:ms], q["A"]])
                ok = np.abs(g["x"][0] - r[0]).max() < 1e-9 and np.abs(G.T @ (g["lam"][0] - r[1])).max() < 1e-7
        if not ok:
            mism.append((trial, n, m, ms, ns, flag, r[3], it, r[4]))
    assert not mism, mism[:10]
    assert len(marks) >= 1, "no degenerate branch was taken: the generator parameters no longer reach them"


def test_warm_sequence(oracle, gpu_lib):
    """config 5: setup + cold solve, then f <- f + 0.05 N(0,I): update(v) + solve reusing the LDL' on the device"""
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS["C2"]
    N, T = 32, 5
    q = O.generate_batch(N, n, m, ms, na, seed)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
    models = []
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        models.append(om)
    f = q["f"].copy()
    warm_iters = []
    for t in range(T + 1):
        if t > 0:
            for k in range(N):
                f[k] = f[k] + 0.05 * np.random.default_rng([45, k, t - 1]).standard_normal(n)
                assert models[k].update(O.UPDATE_v, f=f[k]) == 0
            bm.update(f=f)
        g = bm.solve()
        for k in range(N):
            r = models[k].solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (t, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
            assert bits_equal(g["x"][k], r[0]) and bits_equal(g["lam"][k], r[1])
        if t > 0:
            warm_iters.append(g["iter"].mean())
    assert np.mean(warm_iters) < 15   # far fewer than the ~46 cold iterations
    bm.close()


@pytest.mark.parametrize("eager", [False, True])
@pytest.mark.parametrize("shape", [(13, 40, 5, 5), (12, 48, 12, 6), (20, 40, 0, 8)])
def test_warm_sequence_shapes(oracle, gpu_lib, monkeypatch, shape, eager):
    """update(v) and update(d) (new f, new bounds) on shapes with simple bounds and an odd n; the update is applied inside
    the next solve launch by default (k_ldp_reg mode 2) or by the stand-alone kernel (DAQP_AMD_EAGER_UPDATE=1)"""
    import daqp_amd
    if eager:
        monkeypatch.setenv("DAQP_AMD_EAGER_UPDATE", "1")
    n, m, ms, na = shape
    N, T = 24, 4
    q = O.generate_batch(N, n, m, ms, na, 700 + n)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
    models = []
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        models.append(om)
    f, bu, bl = q["f"].copy(), q["bupper"].copy(), q["blower"].copy()
    for t in range(T + 1):
        if t > 0:
            for k in range(N):
                rng = np.random.default_rng([46, k, t])
                f[k] = f[k] + 0.05 * rng.standard_normal(n)
                if t % 2 == 0:      # every other step the bounds move too (UPDATE_v | UPDATE_d)
                    shift = 0.02 * rng.standard_normal(m)
                    bu[k] = bu[k] + shift; bl[k] = bl[k] + shift
                    assert models[k].update(O.UPDATE_v | O.UPDATE_d, f=f[k], bupper=bu[k], blower=bl[k]) == 0
                else:
                    assert models[k].update(O.UPDATE_v, f=f[k]) == 0
            if t % 2 == 0:
                bm.update(f=f, bupper=bu, blower=bl)
            else:
                bm.update(f=f)
        g = bm.solve()
        for k in range(N):
            r = models[k].solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (t, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
            if r[3] > 0:
                assert bits_equal(g["x"][k], r[0]) and bits_equal(g["lam"][k], r[1])
    bm.close()


def test_model_api(oracle, gpu_lib):
    """reference python tests' Model flow (example_test.py:175-237): setup, solve, update f, update bounds"""
    import daqp_amd
    H, f = np.eye(2), np.array([2.0, 2.0])
    A = np.zeros((0, 2))
    d = daqp_amd.Model()
    flag, _ = d.setup(H, f, A, np.ones(2), -np.ones(2), np.zeros(2, np.int32))
    assert flag >= 0
    x, fval, ef, info = d.solve()
    assert ef == 1 and np.allclose(x, [-1, -1], atol=1e-6)
    assert d.update(f=np.array([-2.0, -2.0])) == 0
    x, _, ef, _ = d.solve()
    assert ef == 1 and np.allclose(x, [1, 1], atol=1e-6)
    d.update(bupper=np.array([0.5, 0.5]), blower=np.array([-0.5, -0.5]))
    x, _, ef, _ = d.solve()
    assert ef == 1 and np.allclose(x, [0.5, 0.5], atol=1e-6)


def test_device_resident_io(oracle, gpu_lib):
    """torch tensors already in HBM are used in place and results can stay on the device"""
    import torch
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS["C2"]
    N = 64
    q = O.generate_batch(N, n, m, ms, na, seed, start=5000)
    dev = {k: torch.from_numpy(q[k]).cuda() for k in ("H", "f", "A", "bupper", "blower")}
    g = daqp_amd.solve_batch(dev["H"], dev["f"], dev["A"], dev["bupper"], dev["blower"], None, ms=ms, out="torch")
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(g["iter"].cpu().numpy(), ref[4])
    assert bits_equal(g["x"].cpu().numpy(), ref[0])


def test_full_size_properties(gpu_lib):
    """BASELINE configs at full size, checked through size-independent properties of the generator:
    every QP optimal, the analytic optimum reproduced, exactly n_active multipliers non-zero,
    KKT stationarity and primal feasibility."""
    import torch
    import daqp_amd
    from daqp_amd.synthetic import generate_batch_torch
    for (N, n, m, ms, na) in ((100_000, 50, 150, 0, 20), (250_000, 12, 48, 12, 6), (10_000, 200, 600, 0, 80)):
        q = generate_batch_torch(N, n, m, ms, na, seed=7)
        g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms, out="torch")
        assert bool((g["exitflag"] == 1).all())
        # tolerances of the reference's own generator tests (core_test.m:16-26: 1e-5): a drawn multiplier
        # ~0 leaves its constraint violated by < primal_tol and therefore (correctly) inactive
        assert float((g["x"] - q["xref"]).abs().max()) < 1e-5
        nact = (g["lam"] != 0).sum(dim=1)
        assert bool((nact <= na).all()) and float((nact != na).double().mean()) < 1e-3
        Afull = torch.cat([torch.eye(n, dtype=torch.float64, device="cuda")[:ms].expand(N, ms, n), q["A"]], dim=1)
        kkt = (q["H"] @ g["x"][:, :, None])[:, :, 0] + q["f"] + (Afull.transpose(1, 2) @ g["lam"][:, :, None])[:, :, 0]
        assert float(kkt.abs().max()) < 1e-5
        ax = (Afull @ g["x"][:, :, None])[:, :, 0]
        assert bool((ax <= q["bupper"] + 1e-5).all()) and bool((ax >= q["blower"] - 1e-5).all())
        del q, g, Afull, kkt, ax
        torch.cuda.empty_cache()


def test_full_size_warm_sequence(gpu_lib):
    """config C5 at full size (100 000 QPs x 10 warm steps, f <- f + 0.05 N(0, I), factors and working sets kept on the
    device): every step every QP optimal, KKT stationarity and primal feasibility for that step's f, and the warm
    solves far cheaper than the cold one"""
    import torch
    import daqp_amd
    from daqp_amd.synthetic import generate_batch_torch
    N, n, m, ms, na, T = 100_000, 50, 150, 0, 20, 10
    q = generate_batch_torch(N, n, m, ms, na, seed=11)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=daqp_amd.UPDATE_unconstrained)
    g = bm.solve(out="torch")
    assert bool((g["exitflag"] == 1).all())
    cold = float(g["iter"].double().mean())
    gen = torch.Generator(device="cuda").manual_seed(5)
    f = q["f"].clone()
    warm = []
    for t in range(T):
        f = f + 0.05 * torch.randn(f.shape, dtype=torch.float64, device="cuda", generator=gen)
        bm.update(f=f)
        g = bm.solve(out="torch")
        assert bool((g["exitflag"] == 1).all()), t
        warm.append(float(g["iter"].double().mean()))
        kkt = (q["H"] @ g["x"][:, :, None])[:, :, 0] + f + (q["A"].transpose(1, 2) @ g["lam"][:, :, None])[:, :, 0]
        assert float(kkt.abs().max()) < 1e-5, t
        ax = (q["A"] @ g["x"][:, :, None])[:, :, 0]
        assert bool((ax <= q["bupper"] + 1e-5).all()) and bool((ax >= q["blower"] - 1e-5).all()), t
        # complementarity: a non-zero multiplier sits on its bound
        onb = torch.minimum((ax - q["bupper"]).abs(), (ax - q["blower"]).abs())
        assert float((onb * (g["lam"] != 0)).max()) < 1e-5, t
    assert max(warm) < 0.5 * cold, (cold, warm)
    bm.close()


# ---- the register kernel's hand-over (n = 64: up to n + 1 = 65 working-set rows, one more than a wavefront has lanes)
@pytest.mark.parametrize("shape", [(64, 128, 0, 20), (64, 100, 0, 63), (64, 65, 0, 60), (64, 128, 8, 56), (63, 128, 0, 62)])
def test_register_kernel_at_64_variables(oracle, gpu_lib, shape):
    """n = 64 runs k_ldp_reg<2,32,*>; a problem whose working set is full (64 rows) when another constraint comes in is flagged and
    redone by the one-wave generic kernel from its untouched state (api.c:305-313: the reference allocates n + 1 rows)"""
    n, m, ms, na = shape
    check_batch(oracle, (n, m, ms, na, 3100 + n + m, 0), 24)


@pytest.mark.parametrize("rows", [5, 14, 24])
def test_register_kernel_forced_hand_over(oracle, gpu_lib, monkeypatch, rows):
    """DAQP_AMD_REG_ROWS caps the rows k_ldp_reg<2,32,*> may hold: working sets of this shape peak at 20-30 rows, so all (5), most (14) or
    some (24) problems are handed to k_ldp in the middle of their solve -- results unchanged, bit for bit"""
    monkeypatch.setenv("DAQP_AMD_REG_ROWS", str(rows))
    check_batch(oracle, (56, 120, 4, 20, 3300, 0), 32)


def test_register_kernel_hand_over_event_trace_and_warm_sequence(oracle, gpu_lib, monkeypatch):
    """warm updates of f and of the bounds across the two kernels: a stored iterate written by k_ldp (more rows than the cap) is k_ldp's
    problem again at the next solve, one written by the register kernel may be handed over later -- add/remove sequences of the
    reference step for step, x / lam / fval bit for bit"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_REG_ROWS", "24")
    n, m, ms, na = 56, 120, 4, 20
    N, T = 16, 4
    q = O.generate_batch(N, n, m, ms, na, 3400)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.enable_trace(4096)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
    models = []
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        om.enable_trace()
        models.append(om)
    f, bu, bl = q["f"].copy(), q["bupper"].copy(), q["blower"].copy()
    nact = []
    for t in range(T + 1):
        if t > 0:
            for k in range(N):
                rng = np.random.default_rng([51, k, t])
                f[k] = f[k] + 0.05 * rng.standard_normal(n)
                shift = 0.02 * rng.standard_normal(m)
                bu[k] = bu[k] + shift; bl[k] = bl[k] + shift
                assert models[k].update(O.UPDATE_v | O.UPDATE_d, f=f[k], bupper=bu[k], blower=bl[k]) == 0
                models[k].enable_trace()
            bm.update(f=f, bupper=bu, blower=bl)
        g = bm.solve()
        tr = bm.read_trace(marks=True)
        for k in range(N):
            r = models[k].solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (t, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
            assert np.array_equal(tr[k], models[k].get_trace(marks=True)), (t, k)
            assert bits_equal(g["x"][k], r[0]) and bits_equal(g["lam"][k], r[1]) and g["fval"][k] == r[2]
        nact.append((g["lam"] != 0).sum(axis=1))
    nact = np.array(nact)
    assert (nact > 24).any() and (nact <= 24).any()     # iterates stored by k_ldp (beyond the cap: its problem again at the next solve) and within the cap
    bm.close()


@pytest.mark.parametrize("shape", [(8, 150, 0, 3), (16, 192, 4, 6), (12, 130, 12, 5), (15, 160, 0, 14), (2, 129, 0, 1), (16, 129, 16, 8),
                                   (8, 256, 0, 3), (16, 193, 4, 6), (12, 250, 12, 5), (15, 200, 0, 14), (3, 256, 0, 1)])      # (the second row: k_ldp_reg<4, 8, *>, 193 .. 256 rows)
def test_register_shape_few_variables_many_rows(oracle, gpu_lib, shape):
    """k_ldp_reg<3, 8, *>: n <= 16 with 129 .. 192 rows (three row blocks, eight column pairs) at two waves per SIMD"""
    n, m, ms, na = shape
    check_batch(oracle, (n, m, ms, na, 4100 + n + m, 0), 64)


@pytest.mark.parametrize("shape", [(40, 64, 0, 13), (50, 64, 6, 16), (33, 34, 0, 30), (45, 60, 45, 12), (50, 51, 0, 49)])
def test_register_shape_one_row_block_up_to_50_variables(oracle, gpu_lib, shape):
    """k_ldp_reg<1, 25, *>: 33 <= n <= 50 with at most 64 rows (one row block, 25 column pairs) at two waves per SIMD"""
    n, m, ms, na = shape
    check_batch(oracle, (n, m, ms, na, 4300 + n + m, 0), 64)
