"""The two branches of daqp_ldp that ordinary data never reaches, forced through settings in every device state machine:

  * the refactor repair at a KKT point (reference src/daqp.c:33-46): `refactor_tol = 10` makes every pivot "too small", so the
    first optimum resets the working set, fixes LOWER / UPPER from the sign of lam and re-activates in index order;
  * the cycle guard (reference src/daqp.c:66-85): `progress_tol = 1e30, cycle_tol = 0` calls every add "no progress": one
    rebuild of the factor, then exit flag -2.

Each runs in the register kernel (k_ldp_reg), the generic one-wave kernel (DAQP_AMD_STREAM_M=1: k_ldp), the workgroup
kernel (n >= 65: k_ldp_wg), in both arithmetic modes.  Bar: exit flag,
iteration count and the add / remove / branch-marker trace equal to the oracle's (itself pinned on these settings against the
reference library: oracle/pin_oracle.py), x and lam bit-identical in the exact mode and within 1e-9 in the default one,
and the marker of the branch present in every problem's trace.

Plus the reference's own known-answer test (interfaces/daqp-julia/test/core_tests.jl:19-30): 100 QPs of (n, m, ms, nAct) =
(100, 500, 50, 80) from the generator with an analytic optimum, |x - xref| < 1e-4.
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
XTOL = 1e-9

FAMILIES = {   # name -> (environment, (n, m, ms, n_active), problems)
    "register": ({}, (20, 40, 0, 8), 20),
    "register_c2": ({}, (50, 150, 0, 20), 6),
    "register_c3": ({}, (12, 48, 12, 6), 20),
    "generic": ({"DAQP_AMD_STREAM_M": "1"}, (20, 40, 0, 8), 20),
    "generic_spill": ({"DAQP_AMD_STREAM_M": "1", "DAQP_AMD_FORCE_SPILL": "1"}, (24, 60, 6, 8), 8),
    "workgroup": ({}, (70, 160, 5, 25), 8),
    "workgroup_chains": ({"DAQP_AMD_WG_INVERSE": "0"}, (70, 160, 5, 25), 4),
    "workgroup_c4": ({}, (130, 300, 0, 50), 3),
}
FORCED = {
    "refactor": (dict(refactor_tol=10.0), "TRACE_REFACTOR", None),
    "cycle": (dict(progress_tol=1e30, cycle_tol=0), "TRACE_CYCLE_RESET", -2),
}


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float64).view(np.uint64),
                          np.ascontiguousarray(b, np.float64).view(np.uint64))


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "default"])
@pytest.mark.parametrize("branch", sorted(FORCED))
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_forced_branch(oracle, gpu_lib, monkeypatch, family, branch, exact):
    import daqp_amd
    from daqp_amd import api
    env, (n, m, ms, na), N = FAMILIES[family]
    kw, marker_name, want_flag = FORCED[branch]
    marker = getattr(api, marker_name)
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    monkeypatch.setenv("DAQP_AMD_NO_RECHECK", "1")     # (no problem here is infeasible: the default-mode kernels are judged unassisted)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    q = O.generate_batch(N, n, m, ms, na, 4242 + n, start=100)
    bm = daqp_amd.BatchModel(N, n, m, ms, **kw)
    bm.enable_trace(16384)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
    assert (bm.setup_flags() == 1).all()
    g = bm.solve()
    traces = bm.read_trace(marks=True)
    bm.close()
    hits = 0
    for k in range(N):
        om = oracle.model(n, m, ms, 0, settings=O.default_settings(**kw))
        assert om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None) >= 0
        om.enable_trace()
        r = om.solve()
        ot = om.get_trace(marks=True)
        assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (family, branch, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
        if want_flag is not None:
            assert r[3] == want_flag, (k, r[3])
        assert np.array_equal(traces[k], ot), (family, branch, k)
        hits += int((traces[k] == marker).sum())
        assert (ot == marker).sum() >= 1, (family, branch, k, "the oracle did not take the branch: the test would prove nothing")
        if r[3] > 0:
            if exact:
                assert bits_equal(g["x"][k], r[0]) and bits_equal(g["lam"][k], r[1]) and g["fval"][k] == r[2], (family, branch, k)
            else:
                assert np.array_equal(np.sign(g["lam"][k]), np.sign(r[1])), (family, branch, k)
                assert np.abs(g["x"][k] - r[0]).max() < XTOL * max(1.0, np.abs(r[0]).max()), (family, branch, k)
    assert hits >= N, (family, branch, hits)


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "default"])
def test_reference_known_answer_100_500(oracle, gpu_lib, monkeypatch, exact):
    """core_tests.jl:19-30: nQPs = 100, n = 100, m = 500, ms = 50, nAct = 80, kappa = 1e2; x within 1e-4 of the generator's
    analytic optimum (workgroup kernel with simple bounds; also against the oracle at the north_star bar)"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    n, m, ms, na = 100, 500, 50, 80
    N = 100
    q = O.generate_batch(N, n, m, ms, na, 1900)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert (g["exitflag"] == 1).all()
    err = np.abs(g["x"] - q["xref"]).max(axis=1)
    assert err.max() < 1e-4, err.max()
    ref = oracle.quadprog_batch(q["H"][:24], q["f"][:24], q["A"][:24], q["bupper"][:24], q["blower"][:24], None, ms=ms)
    assert np.array_equal(g["exitflag"][:24], ref[3]) and np.array_equal(g["iter"][:24], ref[4])
    assert np.array_equal(np.sign(g["lam"][:24]), np.sign(ref[1]))
    if exact:
        assert bits_equal(g["x"][:24], ref[0]) and bits_equal(g["lam"][:24], ref[1])
    else:
        assert np.abs(g["x"][:24] - ref[0]).max() < XTOL * max(1.0, np.abs(ref[0]).max())


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "default"])
@pytest.mark.parametrize("shape,N", [((500, 2500, 250, 400), 8), ((300, 700, 20, 120), 6), ((257, 520, 0, 90), 4)])
def test_beyond_256_rows(oracle, gpu_lib, monkeypatch, shape, N, exact):
    """working sets of more than 256 rows -- the top of the reference's own benchmark ladder, (n, m, ms, nAct) = (500, 2500, 250, 400)
    (interfaces/daqp-julia/test/benchmark.jl:38): generic setup kernel with eight 64-column blocks per lane, one-wave solve
    kernel with eight 64-row chunks, factors and active-row cache in HBM scratch.  Parity, not speed."""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    n, m, ms, na = shape
    q = O.generate_batch(N, n, m, ms, na, 3100 + n)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert (ref[3] == 1).all()
    assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4]), (g["exitflag"], ref[3], g["iter"], ref[4])
    assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1]))
    assert np.abs(g["x"] - q["xref"]).max() < 1e-4
    if exact:
        assert bits_equal(g["x"], ref[0]) and bits_equal(g["lam"], ref[1])
    else:
        assert np.abs(g["x"] - ref[0]).max() < XTOL * max(1.0, np.abs(ref[0]).max())


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "default"])
def test_beyond_256_rows_with_equalities_and_soft_rows(oracle, gpu_lib, monkeypatch, exact):
    """n = 258 with equalities (sense 5) and soft rows (sense 8): the eight-chunk one-wave kernel's activation pass (equalities
    enter the working set at setup) and its soft-constraint branches, working-set capacity n + n_soft + 1 > 256"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    n, m, ms, na = 258, 520, 6, 70
    N = 4
    qs = []
    for k in range(N):
        q = O.generate_qp(n, m, ms, na, rng=[3300, k])
        q = {key: q[key] for key in ("H", "f", "A", "bupper", "blower", "sense")}
        qs.append(O.add_sense_variety(q, ms, 3, 2, [3301, k]))
    b = {key: np.stack([q[key] for q in qs]) for key in ("H", "f", "A", "bupper", "blower", "sense")}
    g = daqp_amd.solve_batch(b["H"], b["f"], b["A"], b["bupper"], b["blower"], b["sense"], ms=ms)
    for k in range(N):
        r = oracle.quadprog(qs[k]["H"], qs[k]["f"], qs[k]["A"], qs[k]["bupper"], qs[k]["blower"], qs[k]["sense"])
        assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (k, g["exitflag"][k], r[3], g["iter"][k], r[4])
        if r[3] > 0:
            assert np.array_equal(np.sign(g["lam"][k]), np.sign(r[1])), k
            if exact:
                assert bits_equal(g["x"][k], r[0]) and bits_equal(g["lam"][k], r[1]) and g["fval"][k] == r[2], k
            else:
                assert np.abs(g["x"][k] - r[0]).max() < XTOL * max(1.0, np.abs(r[0]).max()), k
