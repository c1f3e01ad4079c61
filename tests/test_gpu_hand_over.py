"""The register kernel's hand-over (n = 64: the reference's working set holds n + 1 = 65 rows, api.c:305-313; a wavefront has 64 lanes) on
the entry points the batch tests of test_gpu_parity.py / test_gpu_fast_mode.py do not go through: the single-problem drop-in symbols
(setup_daqp / daqp_update_ldp / daqp_solve, daqp_quadprog) under every update mask, BatchModel.update under every mask, the sharded
entry.  Checked against the pinned oracle: exact mode bit for bit, default mode at the north_star bar (flags, iteration counts,
active sets identical, |dx| < 1e-9)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
XTOL = 1e-9
R, M_, V, D, S = O.UPDATE_Rinv, O.UPDATE_M, O.UPDATE_v, O.UPDATE_d, O.UPDATE_sense
# (shape, DAQP_AMD_REG_ROWS or 0): n = 64 as it is; n = 56 with the cap forced below its working sets (20-30 rows)
CASES = [((64, 128, 0, 24), 0), ((64, 100, 6, 60), 0), ((56, 120, 4, 20), 12), ((56, 120, 4, 20), 24)]
MASKS = [V | D, M_ | D, D, R, V, S, M_, R | M_ | V | D | S, D | S, V | D]


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float64).view(np.uint64), np.ascontiguousarray(b, np.float64).view(np.uint64))


def compare(tag, exact, g, r):
    """g, r: (x, lam, fval, flag, iter)"""
    assert g[3] == r[3] and g[4] == r[4], (tag, g[3], r[3], g[4], r[4])
    if exact:
        assert bits_equal(g[0], r[0]) and bits_equal(g[1], r[1]) and g[2] == r[2], tag
    else:
        assert np.array_equal(np.sign(g[1]), np.sign(r[1])), tag
        if r[3] > 0:      # (the iterate an INFEASIBLE exit leaves behind is no solution: flag, iteration count and working set are the verdict)
            assert np.abs(g[0] - r[0]).max() < XTOL, tag


def perturbed(q, mask, rng, n, m, ms):
    """new arrays for the bits of `mask` (small moves: the optimum stays where working sets of the same size are)"""
    kw = {}
    if mask & R:
        P = 0.005 * rng.standard_normal((n, n))
        kw["H"] = q["H"] + P @ P.T
    if mask & M_:
        kw["A"] = q["A"] * (1.0 + 1e-4 * rng.standard_normal(q["A"].shape))
    if mask & V:
        kw["f"] = q["f"] + 0.02 * rng.standard_normal(n)
    if mask & D:
        shift = 0.005 * rng.standard_normal(m)
        kw["bupper"] = q["bupper"] + shift; kw["blower"] = q["blower"] + shift
    if mask & S:
        kw["sense"] = np.zeros(m, np.int32)
    return kw


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("case", CASES)
def test_single_problem_symbols_every_mask(oracle, gpu_lib, monkeypatch, case, exact):
    """setup_daqp, then daqp_update_ldp(mask) + daqp_solve for every kind of mask (utils.c:58-221), one workspace"""
    import daqp_amd
    (n, m, ms, na), rows = case
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    monkeypatch.setenv("DAQP_AMD_NO_RECHECK", "1")
    if rows:
        monkeypatch.setenv("DAQP_AMD_REG_ROWS", str(rows))
    for trial in range(3):
        q = O.generate_qp(n, m, ms, na, rng=[3700 + n, trial])
        mdl = daqp_amd.Model()
        flag, _ = mdl.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        om = oracle.model(n, m, ms)
        assert om.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"]) == flag == 1
        cur = dict(q)
        for step, mask in enumerate([0] + MASKS):
            if mask:
                kw = perturbed(cur, mask, np.random.default_rng([53, trial, step]), n, m, ms)
                cur.update(kw)
                assert mdl.update_mask(mask, **kw) == om.update(mask, **kw) == 0, (trial, step, mask)
            x, fval, gflag, info = mdl.solve()
            r = om.solve()
            compare((case, trial, step, mask), exact, (x, info["lam"], fval, gflag, info["iterations"]), r)
        # and daqp_quadprog on the last problem: a fresh workspace, setup + solve in one call
        x, fval, gflag, info = daqp_amd.solve(cur["H"], cur["f"], cur["A"], cur["bupper"], cur["blower"], cur["sense"])
        r = oracle.quadprog(cur["H"], cur["f"], cur["A"], cur["bupper"], cur["blower"], cur["sense"])
        compare((case, trial, "quadprog"), exact, (x, info["lam"], fval, gflag, info["iterations"]), r)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("case", CASES)
def test_batch_update_every_mask(oracle, gpu_lib, monkeypatch, case, exact):
    """daqp_batch_update(mask) + daqp_batch_solve: the deferred / fused v|d update of the register shapes is off for these batches
    (the eager kernels run), every other mask goes through the partial setup kernels"""
    import daqp_amd
    (n, m, ms, na), rows = case
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    monkeypatch.setenv("DAQP_AMD_NO_RECHECK", "1")
    if rows:
        monkeypatch.setenv("DAQP_AMD_REG_ROWS", str(rows))
    N = 12
    qs = [O.generate_qp(n, m, ms, na, rng=[3800 + n, k]) for k in range(N)]
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.setup(*(np.stack([q[k] for q in qs]) for k in ("H", "f", "A", "bupper", "blower", "sense")))
    oms = []
    for q in qs:
        om = oracle.model(n, m, ms)
        assert om.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"]) == 1
        oms.append(om)
    cur = [dict(q) for q in qs]
    for step, mask in enumerate([0] + MASKS):
        if mask:
            kws = [perturbed(cur[k], mask, np.random.default_rng([54, k, step]), n, m, ms) for k in range(N)]
            for k in range(N):
                cur[k].update(kws[k])
                assert oms[k].update(mask, **kws[k]) == 0
            bm.update(mask=mask, **{key: np.stack([kw[key] for kw in kws]) for key in kws[0]})
        g = bm.solve()
        for k in range(N):
            r = oms[k].solve()
            compare((case, k, step, mask), exact, (g["x"][k], g["lam"][k], g["fval"][k], int(g["exitflag"][k]), int(g["iter"][k])), r)
    bm.close()


def test_sharded_entry(oracle, gpu_lib, monkeypatch):
    """daqp_batch_*_multi_shards with the device listed twice: both shards' register launches are followed by their own k_ldp pass --
    the reference's setup_daqp + daqp_solve results bit for bit (exact mode), and the unsharded batch's"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    monkeypatch.setenv("DAQP_AMD_REG_ROWS", "14")
    n, m, ms, na = 56, 120, 4, 20
    N = 40
    q = O.generate_batch(N, n, m, ms, na, 3900)
    one = daqp_amd.BatchModel(N, n, m, ms)
    one.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
    r1 = one.solve()
    one.close()
    mb = daqp_amd.MultiBatchModel(N, n, m, ms, devices=[0, 0])
    mb.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=0)
    g = mb.solve()
    mb.close()
    for k in range(N):
        om = oracle.model(n, m, ms)
        assert om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None) == 1
        r = om.solve()
        compare(("sharded", k), True, (g["x"][k], g["lam"][k], g["fval"][k], int(g["exitflag"][k]), int(g["iter"][k])), r)
    assert bits_equal(g["x"], r1["x"]) and bits_equal(g["lam"], r1["lam"]) and np.array_equal(g["iter"], r1["iter"])
