"""k_ldp_tiny (a build with -DDAQP_AMD_WITH_TINY -- tools/tinybuild.sh -- and DAQP_AMD_TINY=1; skipped on the default build): the 16-problems-per-wavefront solve kernel for tiny shapes (n <= 12, m <= 48, working sets of at most
13 rows) against the oracle -- persistent waves with retire / refill of finished problems, lockstep passes, four lanes per problem.

exact mode: bit-identical x, lam, fval, iteration count, exit flag and add / remove / branch trace; default mode: identical exit
flag, iteration count and active set, |x - x_oracle| < 1e-9.  Covers: config C3 (simple bounds first: the TRI = 3 instantiation),
generic shapes incl. m and n below the template's (padding rows / columns), equalities, soft rows, pre-activated working sets
(activation launch + warm start from the stored iterate), the degenerate family (pivot_last, singular direction, refine_active),
warm update sequences, infeasible problems, batches that are no multiple of 16 and batches far larger than the persistent grid.
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
XTOL = 1e-9


@pytest.fixture(autouse=True)
def tiny_on(monkeypatch, gpu_lib):
    # the kernel is not part of the default build (slower than the register kernel on the shape it was written for):
    # tools/tinybuild.sh links a library that carries it; load that one with DAQP_AMD_LIBRARY=... to run this file
    if not gpu_lib.daqp_amd_has_tiny():
        pytest.skip("library built without -DDAQP_AMD_WITH_TINY (tools/tinybuild.sh)")
    monkeypatch.setenv("DAQP_AMD_TINY", "1")


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float64).view(np.uint64),
                          np.ascontiguousarray(b, np.float64).view(np.uint64))


def compare(g, ref, exact, tag):
    assert np.array_equal(g["exitflag"], ref[3]), (tag, np.nonzero(g["exitflag"] != ref[3])[0][:8])
    ok = ref[3] > 0
    # (default arithmetic too: infeasible verdicts of a first solve are re-derived in the reference's arithmetic, csrc/recheck.hip.h)
    assert np.array_equal(g["iter"], ref[4]), (tag, np.nonzero(g["iter"] != ref[4])[0][:8])
    if exact:
        assert bits_equal(g["x"][ok], ref[0][ok]) and bits_equal(g["lam"][ok], ref[1][ok]) and np.array_equal(g["fval"][ok], ref[2][ok]), tag
    else:
        assert np.array_equal(np.sign(g["lam"][ok]), np.sign(ref[1][ok])), tag
        assert np.abs(g["x"][ok] - ref[0][ok]).max() < XTOL, tag


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "default"])
@pytest.mark.parametrize("N", [16, 1000, 40003])
def test_c3_batches(oracle, gpu_lib, monkeypatch, exact, N):
    """C3; 40 003 problems: 2 500 waves' worth on a 1 024-wave persistent grid, the last wave ragged"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    n, m, ms, na, seed, _ = O.CONFIGS["C3"]
    q = O.generate_batch(N, n, m, ms, na, seed, start=300000)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    compare(g, ref, exact, ("C3", N))
    assert np.abs(g["x"] - q["xref"]).max() < 1e-5


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "default"])
@pytest.mark.parametrize("shape", [(12, 48, 0, 6), (12, 48, 5, 8), (11, 47, 11, 5), (8, 20, 3, 3), (5, 48, 5, 4), (2, 4, 0, 1), (12, 13, 12, 9), (7, 33, 0, 6), (2, 6, 2, 1)])
def test_shapes(oracle, gpu_lib, monkeypatch, exact, shape):
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    n, m, ms, na = shape
    N = 333
    q = O.generate_batch(N, n, m, ms, na, 5100 + 7 * n + m)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    compare(g, ref, exact, shape)


def _nasty_tiny(trial):
    rng = np.random.default_rng([1299, trial])
    eps = 10.0 ** rng.uniform(-13, -2)
    n = int(rng.integers(3, 11)); m = int(rng.integers(n + 4, min(4 * n, 48) + 1)); ms = int(rng.integers(0, min(n, m // 3) + 1))
    na = int(rng.integers(1, min(n, m - ms)))
    n_soft = int(rng.integers(0, min(3, 12 - n) + 1))
    return O.generate_nasty(n, m, ms, na, eps, rng, n_dup=int(rng.integers(0, 5)), n_eq=int(rng.integers(0, 3)),
                            n_soft=n_soft, dep_eq=bool(rng.integers(0, 2)))


def test_degenerate_family_with_trace(oracle, gpu_lib, monkeypatch):
    """near-duplicate rows, dependent equalities, soft rows (n + n_soft + 1 <= 13): flag, iterations, x, lam bit-identical and the
    trace -- adds, removes, pivot_last / singular direction / refine markers -- equal to the oracle's; the markers do occur"""
    import daqp_amd
    from daqp_amd import api
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    counts = {api.TRACE_PIVOT: 0, api.TRACE_SINGULAR: 0, api.TRACE_REFINE: 0}
    seen = set()
    for trial in range(500):
        q = _nasty_tiny(trial)
        n, m = q["f"].size, q["bupper"].size
        ms = m - q["A"].reshape(-1, n).shape[0]
        ns = int(((q["sense"] & O.SOFT) != 0).sum())
        om = oracle.model(n, m, ms, ns)
        if om.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"]) < 0:
            continue
        om.enable_trace()
        r = om.solve()
        bm = daqp_amd.BatchModel(1, n, m, ms, ns)
        bm.enable_trace(2048)
        bm.setup(q["H"][None], q["f"][None], q["A"].reshape(1, m - ms, n), q["bupper"][None], q["blower"][None], q["sense"][None])
        assert bm.setup_flags()[0] == 1, trial
        g = bm.solve()
        tr = bm.read_trace(marks=True)[0]
        bm.close()
        assert g["exitflag"][0] == r[3] and g["iter"][0] == r[4], (trial, g["exitflag"][0], r[3], g["iter"][0], r[4])
        assert np.array_equal(tr, om.get_trace(marks=True)), trial
        if r[3] > 0:
            assert bits_equal(g["x"][0], r[0]) and bits_equal(g["lam"][0], r[1]), trial
        seen.add(int(r[3]))
        for k in counts:
            counts[k] += int((tr == k).sum())
    assert counts[api.TRACE_SINGULAR] > 20 and counts[api.TRACE_REFINE] >= 1, counts
    assert {1, -1} <= seen


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "default"])
def test_sense_variety_batch(oracle, gpu_lib, monkeypatch, exact):
    """equalities (sense 5), soft rows (8), pre-activated rows (1 / 3) in one batch: the activation launch builds the working sets,
    the solve launch starts from the stored iterate"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    n, m, ms, na = 9, 30, 4, 4
    N = 200
    q = O.generate_batch(N, n, m, ms, na, 8801)
    rng = np.random.default_rng(8802)
    sense = np.zeros((N, m), np.int32)
    bu, bl = q["bupper"].copy(), q["blower"].copy()
    for k in range(N):
        rows = rng.permutation(m)
        e = rows[0]
        sense[k, e] = 5; bl[k, e] = bu[k, e]
        for s_ in rows[1:1 + int(rng.integers(0, 3))]:
            sense[k, s_] = 8
        for a_ in rows[3:3 + int(rng.integers(0, 3))]:
            sense[k, a_] = 1 if rng.random() < 0.5 else 3
    ns = int(((sense & 8) != 0).sum(axis=1).max())
    st = O.default_settings()
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], bu, bl, sense, settings=st, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], bu, bl, sense, ms=ms)
    assert n + ns + 1 <= 13
    compare(g, ref, exact, "sense variety")
    assert (ref[3] > 0).sum() > N // 2


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "default"])
def test_warm_sequence(oracle, gpu_lib, monkeypatch, exact):
    """setup -> solve -> {update(f) -> solve}*: the persistent iterate (L with its diagonal slots, vectors, WS, sense) written by one
    launch is what the next one resumes from"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    n, m, ms, na, seed, _ = O.CONFIGS["C3"]
    N, T = 96, 4
    q = O.generate_batch(N, n, m, ms, na, seed, start=5000)
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
                f[k] = f[k] + 0.05 * np.random.default_rng([46, k, t]).standard_normal(n)
                models[k].update(O.UPDATE_v, f=f[k])
            bm.update(f=f)
        g = bm.solve()
        na_g, ws_g = bm.working_sets()
        for k in range(N):
            r = models[k].solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (t, k)
            if exact:
                assert bits_equal(g["x"][k], r[0]) and bits_equal(g["lam"][k], r[1]), (t, k)
            else:
                assert np.array_equal(np.sign(g["lam"][k]), np.sign(r[1])) and np.abs(g["x"][k] - r[0]).max() < XTOL, (t, k)
            ws_o = models[k].state()[0]
            assert na_g[k] == ws_o.size and np.array_equal(ws_g[k, : na_g[k]], ws_o), (t, k)
    bm.close()


def test_same_results_as_the_register_kernel(oracle, gpu_lib, monkeypatch):
    """exact mode: k_ldp_tiny and k_ldp_reg<1, 8> return the same bits"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms, na, seed, _ = O.CONFIGS["C3"]
    N = 64
    q = O.generate_batch(N, n, m, ms, na, seed, start=9000)
    res = {}
    for first in ("1", "0"):
        monkeypatch.setenv("DAQP_AMD_TINY", first)
        bm = daqp_amd.BatchModel(N, n, m, ms)
        bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
        g0 = bm.solve()
        res[first] = g0
        bm.close()
    assert bits_equal(res["1"]["x"], res["0"]["x"]) and np.array_equal(res["1"]["iter"], res["0"]["iter"])


def test_single_problem_entry_points_and_proximal_loop(oracle, gpu_lib, monkeypatch):
    """with the kernel opted in process-wide, the single-problem drop-in path (host mirrors of work->lam_star, daqp_extract_result)
    and the proximal loop around tiny shapes still return the reference's bits: the stored iterate holds lam* scaled by
    ldp2qp_solution (daqp.c:136-138), as the other kernels leave it.  (The whole GPU suite passes under DAQP_AMD_TINY=1.)"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms = 9, 20, 3
    q = O.generate_qp(n, m, ms, 4, rng=[661, 1])
    x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    r = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    assert flag == r[3] and info["iterations"] == r[4] and bits_equal(x, r[0]) and bits_equal(info["lam"], r[1])
    qs = O.generate_singular_qp(n, m, ms, rank=4, rng=[662, 2])
    x, fval, flag, info = daqp_amd.solve(qs["H"], qs["f"], qs["A"], qs["bupper"], qs["blower"], qs["sense"])
    r = oracle.quadprog(qs["H"], qs["f"], qs["A"], qs["bupper"], qs["blower"], qs["sense"])
    assert flag == r[3] and info["iterations"] == r[4] and bits_equal(x, r[0]) and bits_equal(info["lam"], r[1]) and fval == r[2]
    lp = O.generate_lp(n, m, ms, [663, 3])
    x, fval, flag, info = daqp_amd.solve(None, lp["f"], lp["A"], lp["bupper"], lp["blower"], lp["sense"])
    r = oracle.quadprog(None, lp["f"], lp["A"], lp["bupper"], lp["blower"], lp["sense"])
    assert flag == r[3] and info["iterations"] == r[4]
    if r[3] > 0:
        assert bits_equal(x, r[0]) and bits_equal(info["lam"], r[1])
