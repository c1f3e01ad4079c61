"""Default mode of the library: M = A R^-1 on the MFMA f64 matrix cores (summation order differs from the
reference, M equal to ~1e-16 relative).  Bar = BASELINE.json north_star: active sets (index and side),
iteration counts and exit flags identical to the reference algorithm, |x - x_ref|_inf < 1e-9."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
XTOL = 1e-9


@pytest.fixture(autouse=True)
def fast_mode(monkeypatch):
    monkeypatch.delenv("DAQP_AMD_EXACT", raising=False)
    # every problem of this file has an optimum: the default-mode kernels are judged on their own, without the second pass in the
    # reference's arithmetic that an INFEASIBLE verdict gets (csrc/recheck.hip.h) -- a kernel that wrongly reported -1 would
    # otherwise come back corrected, and green
    monkeypatch.setenv("DAQP_AMD_NO_RECHECK", "1")


@pytest.mark.parametrize("cfg,N", [("C1", 256), ("C2", 2048), ("C3", 4096), ("C4", 48)])
def test_fast_mode_parity(oracle, gpu_lib, cfg, N):
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS[cfg]
    q = O.generate_batch(N, n, m, ms, na, seed, start=20000)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(g["exitflag"], ref[3])
    assert np.array_equal(g["iter"], ref[4])
    assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1]))
    assert np.abs(g["x"] - ref[0]).max() < XTOL
    assert np.abs(g["lam"] - ref[1]).max() < 1e-8
    assert np.abs(g["fval"] - ref[2]).max() < 1e-8 * max(1.0, np.abs(ref[2]).max())


@pytest.mark.parametrize("shape", [(7, 20, 3, 3), (16, 40, 4, 6), (31, 64, 0, 10), (32, 64, 0, 12), (33, 70, 5, 12),
                                   (48, 100, 0, 16), (56, 120, 0, 20), (57, 120, 0, 20), (64, 128, 0, 24),
                                   (17, 64, 0, 8), (21, 33, 4, 7), (25, 64, 3, 9), (26, 60, 0, 10), (26, 64, 26, 10),   # (these five: k_ldp_reg<1, 13, true>)
                                   (8, 150, 0, 3), (16, 192, 4, 6), (12, 130, 12, 5), (15, 160, 0, 14),   # (k_ldp_reg<3, 8, true>)
                                   (8, 256, 0, 3), (16, 193, 4, 6), (12, 250, 12, 5),                      # (k_ldp_reg<4, 8, true>)
                                   (40, 64, 0, 13), (50, 64, 6, 16), (33, 34, 0, 30), (45, 60, 45, 12)])   # (k_ldp_reg<1, 25, true>)
def test_fast_mode_shapes(oracle, gpu_lib, shape):
    """the MFMA fragment guards of every setup variant (partial k / column tiles) at the north_star bar"""
    import daqp_amd
    n, m, ms, na = shape
    q = O.generate_batch(64, n, m, ms, na, 900 + n)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4])
    assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1]))
    assert np.abs(g["x"] - ref[0]).max() < XTOL


@pytest.mark.parametrize("shape,rows", [((64, 128, 0, 20), 0), ((64, 100, 0, 63), 0), ((64, 128, 8, 64), 0), ((56, 120, 4, 20), 5), ((56, 120, 4, 20), 14),
                                        ((64, 128, 0, 30), 20)])
def test_fast_mode_register_kernel_hand_over(oracle, gpu_lib, monkeypatch, shape, rows):
    """n = 64 on k_ldp_reg<2,32,true> (65 working-set rows possible, 64 lanes): problems that get there -- or, with DAQP_AMD_REG_ROWS, beyond
    `rows` rows -- are redone by the one-wave generic kernel from their untouched state; north_star bar on every problem"""
    import daqp_amd
    if rows:
        monkeypatch.setenv("DAQP_AMD_REG_ROWS", str(rows))
    n, m, ms, na = shape
    q = O.generate_batch(48, n, m, ms, na, 3600 + n + m)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4])
    assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1]))
    assert np.abs(g["x"] - ref[0]).max() < XTOL


@pytest.mark.parametrize("n,ms", [(17, 17), (20, 3), (25, 25), (33, 1), (36, 36), (41, 20), (49, 7), (50, 50), (52, 30), (57, 57), (61, 16), (63, 63), (64, 40)])
def test_blocked_setup_with_simple_bounds(oracle, gpu_lib, monkeypatch, n, ms):
    """k_setup_blk with simple bounds (round 6): rows < ms of the LDP are the rows of R^-1, normalised (utils.c:569-585), their scaling, their d by
    both routes (utils.c:499-544 and the unconstrained shortcut's 664-676), the packed R^-1 with those rows normalised as the reference keeps it --
    the LDP against the oracle's, the solve at the north_star bar, and the same problems through k_setup_fast (DAQP_AMD_NO_BLK_BOUNDS=1)."""
    import daqp_amd
    m, na = ms + n + 9 + (n % 5), max(2, n // 3)
    N = 16
    q = O.generate_batch(N, n, m, ms, na, 7400 + n + ms)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
    assert (bm.setup_flags() == 1).all()
    for k in (0, N - 1):
        om = oracle.model(n, m, ms)
        assert om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None) == 1
        for name, a, b in zip(("M", "Rinv", "v", "dupper", "dlower", "scaling"), bm.read_ldp(k), om.ldp()):
            assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), (n, ms, k, name, np.abs(a - b).max())
    g = bm.solve()
    bm.close()
    assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4])
    assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1])) and np.abs(g["x"] - ref[0]).max() < XTOL
    g1 = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)      # the unconstrained shortcut's d
    assert np.array_equal(g1["iter"], ref[4]) and np.abs(g1["x"] - ref[0]).max() < XTOL
    monkeypatch.setenv("DAQP_AMD_NO_BLK_BOUNDS", "1")
    g2 = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(g2["iter"], ref[4]) and np.abs(g2["x"] - g1["x"]).max() < 1e-11


def test_blocked_setup_diagonal_hessian_with_bounds_goes_to_the_ordered_kernel(oracle, gpu_lib):
    """a diagonal H with simple bounds is the reference's RinvD branch, which scales the bounds its own way (utils.c:284-312): k_setup_blk hands
    such a problem, untouched, to the ordered kernel behind it -- bit-identical LDP to the oracle's there"""
    import daqp_amd
    n, m, ms, na = 40, 90, 12, 10
    q = O.generate_batch(6, n, m, ms, na, 7600)
    for k in range(0, 6, 2):
        q["H"][k] = np.diag(np.abs(np.diag(q["H"][k])) + 0.5)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4]), (g["exitflag"], ref[3], g["iter"], ref[4])
    ok = ref[3] > 0
    assert np.abs(g["x"][ok] - ref[0][ok]).max() < XTOL and np.array_equal(np.sign(g["lam"][ok]), np.sign(ref[1][ok]))


@pytest.mark.parametrize("n", [17, 19, 20, 21, 25, 32, 33, 35, 36, 37, 41, 48, 49, 50, 51, 52, 53, 55, 56, 57, 61, 64])
def test_blocked_setup_every_block_shape(oracle, gpu_lib, monkeypatch, n):
    """k_setup_blk (csrc/setup_blk.hip.h: 16 < n <= 64, default arithmetic -- Cholesky and inverse as 16 x 16 tiles
    on the matrix cores): every block count, column width and tail width (n mod 16 in 1..4: the last columns on the vector pipe), odd n
    (rows staged through registers) and even n (direct HBM -> LDS copies), row counts that end in a partial row tile; the LDP against
    the oracle's (R^-1, M, v, d to ~1e-13 relative: another summation order), then the solve at the north_star bar; and the same
    problems through the ordered kernel (DAQP_AMD_NO_BLK_SETUP=1) give the same iterations and active sets."""
    import daqp_amd
    m, na = 2 * n + 7 + (n % 5), max(2, n // 3)
    N = 24
    q = O.generate_batch(N, n, m, 0, na, 7000 + n)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=0)
    bm = daqp_amd.BatchModel(N, n, m, 0)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
    assert (bm.setup_flags() == 1).all()
    for k in (0, N - 1):
        om = oracle.model(n, m, 0)
        assert om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None) == 1
        for name, a, b in zip(("M", "Rinv", "v", "dupper", "dlower", "scaling"), bm.read_ldp(k), om.ldp()):
            assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), (n, k, name, np.abs(a - b).max())
    g = bm.solve()
    bm.close()
    assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4])
    assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1]))
    assert np.abs(g["x"] - ref[0]).max() < XTOL
    g1 = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=0)      # the unconstrained shortcut's d (utils.c:664-676)
    assert np.array_equal(g1["iter"], ref[4]) and np.abs(g1["x"] - ref[0]).max() < XTOL


def test_blocked_setup_hands_doubtful_hessians_to_the_ordered_kernel(oracle, gpu_lib):
    """a Hessian that is singular, nearly so (pivot ratio at the threshold of utils.c:354-356) or diagonal inside a batch of ordinary
    ones: k_setup_blk marks what it does not call clearly regular and the ordered kernel behind it decides as the reference does --
    shift + proximal loop, or no shift -- while a diagonal H takes the RinvD branch in k_setup_blk itself"""
    import daqp_amd
    n, m, na, N = 40, 90, 12, 16
    q = O.generate_batch(N, n, m, 0, na, 7700)
    rng = np.random.default_rng(7701)
    for k in (1, 5):        # rank-deficient
        G = rng.standard_normal((n - 6, n))
        q["H"][k] = G.T @ G
    for k, eps in ((2, 3e-11), (6, 0.9e-11), (9, 1.5e-11), (10, 4e-11)):     # smallest pivot / largest pivot around zero_tol = 1e-11
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        ev = np.linspace(1.0, 3.0, n); ev[0] = eps
        q["H"][k] = (Q * ev) @ Q.T
    for k in (3, 12):       # diagonal
        q["H"][k] = np.diag(1.0 + rng.random(n))
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=0)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=0)
    assert np.array_equal(g["exitflag"], ref[3]), (g["exitflag"], ref[3])
    assert np.array_equal(g["iter"], ref[4]), (g["iter"], ref[4])
    ok = ref[3] > 0
    assert np.abs(g["x"][ok] - ref[0][ok]).max() < 1e-7 and np.array_equal(np.sign(g["lam"][ok]), np.sign(ref[1][ok]))
    for k in (3, 12):
        assert np.array_equal(g["x"][k].view(np.uint64), ref[0][k].view(np.uint64)) or np.abs(g["x"][k] - ref[0][k]).max() < XTOL


@pytest.mark.parametrize("shape", [(65, 150, 0, 30), (100, 260, 7, 40), (129, 200, 10, 30), (229, 400, 20, 60), (229, 420, 0, 205), (187, 371, 0, 92), (110, 330, 0, 35), (114, 400, 3, 45)])
def test_fast_mode_workgroup_kernel_shapes(oracle, gpu_lib, shape):
    """the workgroup solve kernel in the default arithmetic: the inverse factor W = L^-1 (CSP / append / delete as matrix-vector
    products over all waves), primal step and Gram column summed in per-wave segments, fp32-screened scan.  The last shape's
    working sets pass 191 rows, where the kernel converts W back to L and continues on the substitution chains."""
    import daqp_amd
    n, m, ms, na = shape
    q = O.generate_batch(12 if na < 100 else 3, n, m, ms, na, 1900 + n + na)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4])
    assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1]))
    assert np.abs(g["x"] - ref[0]).max() < XTOL


def test_fast_mode_ldp_close(oracle, gpu_lib):
    """the default mode's LDP (fused multiply-adds in the Cholesky / inverse sweep, M off the matrix cores) matches the reference's
    to rounding"""
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS["C2"]
    q = O.generate_batch(4, n, m, ms, na, seed)
    bm = daqp_amd.BatchModel(4, n, m, ms)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
    for k in range(4):
        om = oracle.model(n, m, ms)
        om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        M, R, v, du, dl, sc = bm.read_ldp(k)
        Mo, Ro, vo, duo, dlo, sco = om.ldp()
        assert np.abs(R - Ro).max() < 1e-13 * np.abs(Ro).max() and np.abs(v - vo).max() < 1e-13 * np.abs(vo).max()
        assert np.abs(M - Mo).max() < 1e-14 and np.abs(sc - sco).max() < 1e-13 * np.abs(sco).max()
        assert np.abs(du - duo).max() < 1e-12 and np.abs(dl - dlo).max() < 1e-12
    bm.close()


def test_fast_mode_warm_sequence(oracle, gpu_lib):
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS["C2"]
    N, T = 64, 4
    q = O.generate_batch(N, n, m, ms, na, seed, start=777)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
    models = []
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        models.append(om)
    f = q["f"].copy()
    for t in range(T + 1):
        if t > 0:
            for k in range(N):
                f[k] = f[k] + 0.05 * np.random.default_rng([45, k, t - 1]).standard_normal(n)
                models[k].update(O.UPDATE_v, f=f[k])
            bm.update(f=f)
        g = bm.solve()
        for k in range(N):
            r = models[k].solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4]
            assert np.array_equal(np.sign(g["lam"][k]), np.sign(r[1])) and np.abs(g["x"][k] - r[0]).max() < XTOL
    bm.close()


def test_inverse_factor_matches_chains(oracle, gpu_lib, monkeypatch):
    """default mode with and without the inverse factor (DAQP_AMD_WG_INVERSE=0: L and the substitution chains): the same add /
    remove sequence step for step, x and lam equal to rounding -- on a shape with removals in most problems, with a warm
    update afterwards (the stored iterate is always L: the warm solve starts on the chains)"""
    import daqp_amd
    n, m, ms, na = 90, 220, 6, 35
    N = 16
    q = O.generate_batch(N, n, m, ms, na, 7700)
    res = {}
    for inv in ("1", "0"):
        monkeypatch.setenv("DAQP_AMD_WG_INVERSE", inv)
        bm = daqp_amd.BatchModel(N, n, m, ms)
        bm.enable_trace(8192)
        bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=daqp_amd.UPDATE_unconstrained)
        g = bm.solve()
        tr = bm.read_trace()
        bm.update(f=q["f"] * 1.05)
        g2 = bm.solve()
        res[inv] = (g, tr, g2)
        bm.close()
    a, b = res["1"], res["0"]
    assert np.array_equal(a[0]["iter"], b[0]["iter"]) and np.array_equal(a[0]["exitflag"], b[0]["exitflag"])
    assert all(np.array_equal(x, y) for x, y in zip(a[1], b[1])), "add / remove sequences differ"
    assert any((np.asarray(t) < 0).any() for t in a[1]), "no removal in the sample: the delete sweep was not exercised"
    assert np.abs(a[0]["x"] - b[0]["x"]).max() < 1e-11 and np.abs(a[0]["lam"] - b[0]["lam"]).max() < 1e-9
    assert np.array_equal(a[2]["iter"], b[2]["iter"]) and np.abs(a[2]["x"] - b[2]["x"]).max() < 1e-11
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(a[0]["iter"], ref[4]) and np.abs(a[0]["x"] - ref[0]).max() < XTOL


def test_fast_mode_shared_structure_large(oracle, gpu_lib):
    """shared-structure batch at n = 80 in the default arithmetic: generic setup on the matrix cores, workgroup kernel with the
    fp32-screened scan and the inverse factor on the first (cold) solve, warm updates afterwards on the stored L"""
    import daqp_amd
    n, m, ms, na = 80, 200, 5, 30
    N = 24
    q0 = O.generate_qp(n, m, ms, na, rng=[821, n])
    rng = np.random.default_rng([822, n])
    # (small perturbations of the generator's QP: its optimum stays well conditioned -- large ones produce multipliers of 1e4
    #  and an x that moves by 1e-7 relative under ANY change of rounding, which says nothing about the kernels)
    f = q0["f"][None, :] + 0.02 * rng.standard_normal((N, n))
    shift = 0.005 * rng.standard_normal((N, m))
    bu, bl = q0["bupper"][None, :] + shift, q0["blower"][None, :] + shift
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.setup_shared(q0["H"], f, q0["A"], bu, bl, None)
    models = []
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(q0["H"], f[k], q0["A"], np.full(m, 1e30), np.full(m, -1e30), None)
        assert om.update(O.UPDATE_v | O.UPDATE_d, f=f[k], bupper=bu[k], blower=bl[k]) == 0
        models.append(om)
    for t in range(3):
        if t > 0:
            f = f + 0.01 * rng.standard_normal((N, n))
            bm.update(f=f)
            for k in range(N):
                assert models[k].update(O.UPDATE_v, f=f[k]) == 0
        g = bm.solve()
        for k in range(N):
            r = models[k].solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (t, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
            assert np.array_equal(np.sign(g["lam"][k]), np.sign(r[1])), (t, k)
            assert np.abs(g["x"][k] - r[0]).max() < XTOL * max(1.0, np.abs(r[0]).max(), np.abs(r[1]).max()), (t, k, np.abs(g["x"][k] - r[0]).max(), np.abs(r[1]).max())
    bm.close()


@pytest.mark.parametrize("shape", [(200, 600, 0, 80), (70, 150, 6, 20), (65, 130, 65, 10), (129, 333, 10, 40), (208, 250, 3, 30), (100, 64, 0, 10), (90, 40, 9, 8)])
def test_general_rows_as_their_own_launch(oracle, gpu_lib, monkeypatch, shape):
    """k_setup_m (csrc/setup_m.hip.h): the generic setup's general rows -- M = A R^-1 on the matrix cores with R^-1 shared through LDS by
    the four waves of a workgroup, normalisation, d, both images -- as a launch of their own behind k_setup.  The LDP it leaves
    equals the reference's to rounding and the one-kernel path's (DAQP_AMD_NO_SETUP_M=1) to rounding, for row counts that are no
    multiple of 64 or 16, simple bounds in front (image blocks that straddle workgroups), a zero row (IMMUTABLE), the setup_daqp and the
    daqp_quadprog variants of d; the solves that follow take the reference's path."""
    import daqp_amd
    n, m, ms, na = shape
    N = 5
    q = O.generate_batch(N, n, m, ms, na, 3100 + n + m)
    if m - ms > 3:
        q["A"][1, 2] = 0.0                                # a zero row with 0 inside its bounds: IMMUTABLE, scaling 1
        q["bupper"][1, ms + 2] = 1.0; q["blower"][1, ms + 2] = -1.0
    ldps = {}
    for env in ("0", "1"):
        monkeypatch.setenv("DAQP_AMD_NO_SETUP_M", env)
        for mask in (0, 64 + 128):
            bm = daqp_amd.BatchModel(N, n, m, ms)
            bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=mask)
            assert (bm.setup_flags() == 1).all()
            ldps[env, mask] = [bm.read_ldp(k) for k in range(N)]
            g = bm.solve()
            if mask:
                ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
                assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4]), (env, mask)
                assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1])) and np.abs(g["x"] - ref[0]).max() < XTOL
            bm.close()
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        Mo, Ro, vo, duo, dlo, sco = om.ldp()
        for env in ("0", "1"):
            M, R, v, du, dl, sc = ldps[env, 0][k]
            assert np.abs(M - Mo).max() < 1e-13 and np.abs(sc - sco).max() < 1e-12 * np.abs(sco).max(), (env, k)
            assert np.abs(du - duo).max() < 1e-11 and np.abs(dl - dlo).max() < 1e-11, (env, k)
        for mask in (0, 192):
            a, b_ = ldps["0", mask][k], ldps["1", mask][k]
            assert np.abs(a[0] - b_[0]).max() < 1e-13 and np.abs(a[3] - b_[3]).max() < 1e-11 and np.abs(a[5] - b_[5]).max() < 1e-12 * np.abs(b_[5]).max(), (mask, k)
    monkeypatch.setenv("DAQP_AMD_NO_SETUP_M", "0")
    # an infeasible zero row, and a problem whose unconstrained optimum is feasible (the shortcut: one iteration, no multipliers)
    if m - ms > 3:
        q2 = {kk: vv.copy() for kk, vv in q.items()}
        q2["A"][3, 1] = 0.0; q2["bupper"][3, ms + 1] = -1.0; q2["blower"][3, ms + 1] = -2.0
        q2["bupper"][4] = 1e6; q2["blower"][4] = -1e6
        g = daqp_amd.solve_batch(q2["H"], q2["f"], q2["A"], q2["bupper"], q2["blower"], None, ms=ms)
        ref = oracle.quadprog_batch(q2["H"], q2["f"], q2["A"], q2["bupper"], q2["blower"], None, ms=ms)
        assert ref[3][3] == -1 and ref[3][4] == 1 and ref[4][4] == 1
        assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4])
        assert np.abs(g["x"][4] - ref[0][4]).max() < XTOL and not g["lam"][4].any()


@pytest.mark.parametrize("shape", [(200, 600, 0, 80), (200, 480, 12, 60), (193, 300, 0, 40), (176, 520, 0, 50), (161, 610, 5, 30),
                                   (100, 300, 0, 40), (65, 150, 0, 25), (128, 300, 7, 40), (81, 200, 3, 30), (134, 330, 0, 40)])   # (the second row: factors that would fit k_setup's LDS -- round 6 gives them the same launch)
def test_factorisation_as_its_own_launch(oracle, gpu_lib, monkeypatch, shape):
    """k_fact_wg (csrc/setup_fact.hip.h): Cholesky factor and inverse of the generic setup for the shapes whose factors do not fit
    LDS twice (the n = 200 class), default arithmetic -- a workgroup per problem, the packed triangle in LDS, sixteen-row panels, the
    rank-16 updates and the inverse's block products on the matrix cores.  R^-1, v, M, d equal the reference's to rounding and k_setup's
    own ordered factorisation (DAQP_AMD_NO_FACT_WG=1) to rounding, for n that is and is not a multiple of sixteen; the solves take the
    reference's path.  What the kernel gives up on goes to k_setup's own code and ends with the reference's verdict: a diagonal
    Hessian (the RinvD branch), an indefinite one (-5 with eps_prox = 0) and a numerically singular one (the regularising passes)."""
    import daqp_amd
    n, m, ms, na = shape
    N = 6
    q = O.generate_batch(N, n, m, ms, na, 4100 + n + m)
    ldps, sols = {}, {}
    for env in (None, "1"):
        if env: monkeypatch.setenv("DAQP_AMD_NO_FACT_WG", env)
        else: monkeypatch.delenv("DAQP_AMD_NO_FACT_WG", raising=False)
        bm = daqp_amd.BatchModel(N, n, m, ms)
        bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=64 + 128)
        assert (bm.setup_flags() == 1).all()
        ldps[env] = [bm.read_ldp(k) for k in range(N)]
        sols[env] = bm.solve()
        bm.close()
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    for env in (None, "1"):
        g = sols[env]
        assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4]), env
        assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1])) and np.abs(g["x"] - ref[0]).max() < XTOL
    differs = False
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None, init_mask=64 + 128)
        Mo, Ro, vo, duo, dlo, sco = om.ldp()
        M, R, v, du, dl, sc = ldps[None][k]
        assert np.abs(R - Ro).max() < 1e-13 * np.abs(Ro).max() and np.abs(v - vo).max() < 1e-12 * max(1.0, np.abs(vo).max()), k
        assert np.abs(M - Mo).max() < 1e-13 and np.abs(sc - sco).max() < 1e-12 * np.abs(sco).max(), k
        assert np.abs(du - duo).max() < 1e-10 and np.abs(dl - dlo).max() < 1e-10, k
        assert np.array_equal(ldps["1"][k][1], Ro), k          # k_setup's own factorisation is the reference's, bit for bit
        differs |= not np.array_equal(R, Ro)
    assert differs                                             # (the new kernel did run: another summation order leaves other last bits)
    monkeypatch.delenv("DAQP_AMD_NO_FACT_WG", raising=False)
    # what the kernel gives up on
    q2 = {kk: vv.copy() for kk, vv in q.items()}
    q2["H"][1] = np.diag(np.diag(q["H"][1]))                     # diagonal
    ev = np.linalg.eigvalsh(q["H"][2])
    q2["H"][2] = q["H"][2] - 1.5 * ev[0] * np.eye(n)             # indefinite: a negative pivot
    U = np.linalg.qr(np.random.default_rng(5).standard_normal((n, n)))[0]
    q2["H"][3] = (U[:, : n - 3] * np.linspace(1.0, 2.0, n - 3)) @ U[:, : n - 3].T    # rank n - 3: the proximal loop
    for eps in (0.0, -1e-6):
        g = daqp_amd.solve_batch(q2["H"], q2["f"], q2["A"], q2["bupper"], q2["blower"], None, ms=ms, eps_prox=eps)
        r = oracle.quadprog_batch(q2["H"], q2["f"], q2["A"], q2["bupper"], q2["blower"], None, settings=O.default_settings(eps_prox=eps), ms=ms)
        assert np.array_equal(g["exitflag"], r[3]), (eps, g["exitflag"], r[3])
        ok = r[3] > 0
        assert np.array_equal(g["iter"][ok], r[4][ok]) and np.abs(g["x"][ok] - r[0][ok]).max() < 1e-7 * max(1.0, np.abs(r[0][ok]).max())


def test_factorisation_launch_on_reused_batches_and_single_problems(oracle, gpu_lib):
    """k_fact_wg behind the other entry points of the n = 200 class: a batch set up twice (new Hessians over the old records), a
    warm update in between, and the single-problem drop-in call (a batch of one)."""
    import daqp_amd
    n, m, ms, na, N = 200, 420, 0, 50, 5
    q1 = O.generate_batch(N, n, m, ms, na, 5151)
    q2 = O.generate_batch(N, n, m, ms, na, 5252)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    for q in (q1, q2):
        bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=0)
        g = bm.solve()
        models = []
        for k in range(N):
            om = oracle.model(n, m, ms)
            om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
            r = om.solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (k, g["iter"][k], r[4])
            assert np.abs(g["x"][k] - r[0]).max() < XTOL * max(1.0, np.abs(r[0]).max())
            models.append(om)
        f2 = q["f"] * 1.05 + 0.01
        bm.update(f=f2)
        g = bm.solve()
        for k in range(N):
            models[k].update(O.UPDATE_v, f=f2[k])
            r = models[k].solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], ("warm", k, g["iter"][k], r[4])
            assert np.abs(g["x"][k] - r[0]).max() < XTOL * max(1.0, np.abs(r[0]).max())
    bm.close()
    x, fval, flag, info = daqp_amd.solve(q1["H"][0], q1["f"][0], q1["A"][0], q1["bupper"][0], q1["blower"][0], np.zeros(m, np.int32))
    r = oracle.quadprog(q1["H"][0], q1["f"][0], q1["A"][0], q1["bupper"][0], q1["blower"][0], None)
    assert flag == r[3] and info["iterations"] == r[4] and np.abs(x - r[0]).max() < XTOL * max(1.0, np.abs(r[0]).max())


def test_shared_setup_after_a_factorised_per_problem_setup(oracle, gpu_lib):
    """a batch of the n = 200 class that was set up per problem (k_fact_wg's records say "factored") and is then set up SHARED with
    another Hessian: the one-problem descriptor of the shared setup must not inherit those records (k_setup would skip its own
    Cholesky / inverse and take the previous problem 0's R^-1 out of the scratch)"""
    import daqp_amd
    n, m, ms, na, N = 200, 300, 0, 30, 4
    q = O.generate_batch(N, n, m, ms, na, 6161)
    q2 = O.generate_batch(1, n, m, ms, na, 6262)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
    bm.solve()
    rng = np.random.default_rng(7)
    bu = q2["bupper"][0] + rng.uniform(0.0, 0.5, (N, m)); bl = q2["blower"][0] - rng.uniform(0.0, 0.5, (N, m))     # (widened: every state feasible)
    fs = q2["f"][0] * (1.0 + 0.1 * rng.standard_normal((N, 1)))
    bm.setup_shared(q2["H"][0], fs, q2["A"][0], bu, bl)
    g = bm.solve()
    for k in range(N):
        om = oracle.model(n, m, ms)
        om.setup(q2["H"][0], fs[k], q2["A"][0], bu[k], bl[k], None)
        r = om.solve()
        assert r[3] == 1 and g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (k, g["exitflag"][k], r[3], g["iter"][k], r[4])
        assert np.abs(g["x"][k] - r[0]).max() < XTOL * max(1.0, np.abs(r[0]).max()), k
    bm.close()
