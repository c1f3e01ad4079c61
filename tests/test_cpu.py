"""CPU-only tests (-m "not gpu"): the oracle against committed golden vectors and analytic optima,
host logic, and that the C-ABI library loads and exports every symbol include/daqp_amd.h declares."""
import ctypes as C
import os
import re

import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_matches_generator_optimum(oracle):
    for cfg in ("C1", "C2", "C3"):
        n, m, ms, na, seed, _ = O.CONFIGS[cfg]
        for k in range(20):
            q = O.generate_qp(n, m, ms, na, rng=[seed, k])
            x, lam, fval, flag, it = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
            assert flag == 1
            assert np.abs(x - q["x"]).max() < 1e-8          # reference tests use 1e-4 (core_tests.jl:26-30)
            assert (np.count_nonzero(lam) == na)
            # KKT stationarity and objective identity (core_test.m:16-26)
            Afull = np.vstack([np.eye(n)[:ms], q["A"]])
            assert np.abs(q["H"] @ x + q["f"] + Afull.T @ lam).max() < 1e-7
            assert abs(0.5 * x @ q["H"] @ x + q["f"] @ x - fval) < 1e-8


def test_oracle_hand_examples(oracle):
    # interfaces/daqp-python/test/example_test.py:175-237
    H, f, A = np.eye(2), np.array([2.0, 2.0]), np.zeros((0, 2))
    x, lam, fval, flag, it = oracle.quadprog(H, f, A, np.ones(2), -np.ones(2), np.zeros(2, np.int32))
    assert flag == 1 and np.allclose(x, [-1, -1], atol=1e-6)
    x, *_ = oracle.quadprog(H, -f, A, np.ones(2), -np.ones(2), np.zeros(2, np.int32))
    assert np.allclose(x, [1, 1], atol=1e-6)
    x, *_ = oracle.quadprog(H, f, A, 0.5 * np.ones(2), -0.5 * np.ones(2), np.zeros(2, np.int32))
    assert np.allclose(x, [-0.5, -0.5], atol=1e-6)
    # iter_limit = 1 -> -4 (core_tests.jl:33-35); crossed bounds -> -1 (core_test.m:212-221)
    q = O.generate_qp(20, 40, 0, 8, rng=[1234, 0])
    assert oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, O.default_settings(iter_limit=1))[3] == -4
    bu = q["bupper"].copy(); bu[3] = q["blower"][3] - 1
    assert oracle.quadprog(q["H"], q["f"], q["A"], bu, q["blower"])[3] == -1


def test_oracle_warm_start_one_iteration(oracle):
    """an exact dual warm start needs exactly 1 iteration (core_tests.jl:520-545)"""
    n, m, ms, na, seed, _ = O.CONFIGS["C1"]
    q = O.generate_qp(n, m, ms, na, rng=[seed, 3])
    x, lam, *_ = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    sense = np.zeros(m, np.int32)
    sense[lam > 1e-12] |= O.ACTIVE
    sense[lam < -1e-12] |= O.ACTIVE + O.LOWER
    x2, lam2, fval2, flag, it = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], sense)
    assert flag == 1 and it == 1 and np.abs(x - x2).max() < 1e-10


def test_oracle_golden_vectors(oracle):
    """fixtures written by tests/golden/make_golden.py from the REFERENCE library (strict build)"""
    path = os.path.join(ROOT, "tests", "golden", "golden_quadprog.npz")
    g = np.load(path, allow_pickle=False)
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) >= 40
    for nm in names:
        get = lambda f: g[f"{nm}/{f}"]
        sense = get("sense") if f"{nm}/sense" in g.files else None
        x, lam, fval, flag, it = oracle.quadprog(get("H"), get("f"), get("A"), get("bupper"), get("blower"), sense)
        assert flag == int(get("exitflag")) and it == int(get("iter")), nm
        if flag > 0:
            assert np.array_equal(x.view(np.uint64), get("x").view(np.uint64)), nm
            assert np.array_equal(lam.view(np.uint64), get("lam").view(np.uint64)), nm
            assert fval == float(get("fval")), nm


def test_oracle_golden_warm_sequence(oracle):
    path = os.path.join(ROOT, "tests", "golden", "golden_warm.npz")
    g = np.load(path, allow_pickle=False)
    n, m, ms = int(g["n"]), int(g["m"]), int(g["ms"])
    om = oracle.model(n, m, ms)
    assert om.setup(g["H"], g["f0"], g["A"], g["bupper"], g["blower"], None) == 1
    for t in range(g["fs"].shape[0]):
        if t > 0:
            assert om.update(O.UPDATE_v, f=g["fs"][t]) == 0
        x, lam, fval, flag, it = om.solve()
        assert flag == int(g["exitflag"][t]) and it == int(g["iter"][t])
        assert np.array_equal(x.view(np.uint64), g["x"][t].view(np.uint64))


def _c4_fixture():
    """tests/golden/golden_c4.npz: inputs (stored as float32: the problems were rounded to fp32-representable values before the
    reference solved them) and the reference's outputs of 4 QPs of config C4 (tests/golden/make_golden_configs.py)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_c4.npz"), allow_pickle=False)
    assert (int(g["n"]), int(g["m"]), int(g["ms"])) == (200, 600, 0)
    qs = [{kk: g[f"{k}/{kk}"].astype(np.float64) for kk in ("H", "f", "A", "bupper", "blower")} for k in range(4)]
    return g, qs


def test_oracle_golden_c4(oracle):
    g, qs = _c4_fixture()
    for k, q in enumerate(qs):
        x, lam, fval, flag, it = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
        assert flag == int(g[f"{k}/exitflag"]) == 1 and it == int(g[f"{k}/iter"]) and fval == float(g[f"{k}/fval"])
        assert np.array_equal(x.view(np.uint64), g[f"{k}/x"].view(np.uint64)) and np.array_equal(lam.view(np.uint64), g[f"{k}/lam"].view(np.uint64))


def test_oracle_golden_warm_sequence_c2(oracle):
    """config C5's shape: 10 warm steps on C2-sized QPs, the reference's own outputs at every step"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_warm_c2.npz"), allow_pickle=False)
    n, m, ms, T = int(g["n"]), int(g["m"]), int(g["ms"]), int(g["T"])
    for k in range(2):
        om = oracle.model(n, m, ms)
        assert om.setup(g[f"{k}/H"], g[f"{k}/fs"][0], g[f"{k}/A"], g[f"{k}/bupper"], g[f"{k}/blower"], None) == 1
        for t in range(T + 1):
            if t > 0:
                assert om.update(O.UPDATE_v, f=g[f"{k}/fs"][t]) == 0
            x, lam, fval, flag, it = om.solve()
            assert flag == int(g[f"{k}/exitflag"][t]) and it == int(g[f"{k}/iter"][t]) and fval == float(g[f"{k}/fval"][t]), (k, t)
            assert np.array_equal(x.view(np.uint64), g[f"{k}/x"][t].view(np.uint64)) and np.array_equal(lam.view(np.uint64), g[f"{k}/lam"][t].view(np.uint64))


def test_oracle_golden_proximal(oracle):
    """singular / forcibly shifted Hessians and LPs through the proximal outer loop: fixtures from the REFERENCE (strict build)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_prox.npz"), allow_pickle=False)
    names = sorted({k.split("/")[0] for k in g.files} - {"warm"})
    assert len(names) >= 40
    flags = set()
    for nm in names:
        get = lambda f: g[f"{nm}/{f}"]
        st = O.default_settings(eps_prox=float(get("eps_prox")), eta_prox=float(get("eta_prox")), iter_limit=int(get("iter_limit")))
        H = get("H") if get("H").size else None    # an LP
        x, lam, fval, flag, it = oracle.quadprog(H, get("f"), get("A"), get("bupper"), get("blower"), get("sense"), settings=st)
        assert flag == int(get("exitflag")), nm
        flags.add(flag)
        if flag != -5:
            assert it == int(get("iter")), nm
        if flag > 0:
            assert np.array_equal(x.view(np.uint64), get("x").view(np.uint64)), nm
            assert np.array_equal(lam.view(np.uint64), get("lam").view(np.uint64)), nm
            assert fval == float(get("fval")), nm
    assert flags == {1, -3, -4, -5}
    n, m, ms = int(g["warm/n"]), int(g["warm/m"]), int(g["warm/ms"])
    om = oracle.model(n, m, ms)
    assert om.setup(g["warm/H"], g["warm/fs"][0], g["warm/A"], g["warm/bupper"], g["warm/blower"], None) == 1
    for t in range(g["warm/fs"].shape[0]):
        if t > 0:
            assert om.update(O.UPDATE_v, f=g["warm/fs"][t]) == 0
        x, lam, fval, flag, it = om.solve()
        assert flag == int(g["warm/exitflag"][t]) and it == int(g["warm/iter"][t]), t
        assert np.array_equal(x.view(np.uint64), g["warm/x"][t].view(np.uint64)) and fval == float(g["warm/fval"][t])
        assert np.array_equal(lam.view(np.uint64), g["warm/lam"][t].view(np.uint64))


def test_oracle_golden_update_masks(oracle):
    """every daqp_update_ldp mask (utils.c:58-221): the oracle replays the sequences the reference library wrote into
    tests/golden/golden_update_masks.npz -- update flags, x, lam, fval, iterations, exit flags and working sets bit for bit"""
    import mask_replay as MR
    steps_checked = 0
    for shape, (n, m, ms) in MR.SHAPES.items():
        for mask, steps in MR.sequences(shape):
            for trial in range(MR.TRIALS):
                q = MR.base(shape, trial)
                om = oracle.model(n, m, ms, ns=MR.NS[shape])
                assert om.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"]) == 1
                for s in range(-1, steps):
                    kw, exp = MR.step(shape, trial, mask, s)
                    uflag = om.update(mask, **kw) if s >= 0 else 0
                    x, lam, fval, flag, it = om.solve()
                    MR.check(f"{shape}/{trial}/mask {mask}/step {s}", dict(x=x, lam=lam, fval=fval, flag=flag, iter=it, uflag=uflag,
                                                                      ws=om.state()[0]), exp, exact=True)
                    steps_checked += 1
    assert steps_checked > 700


def test_c_abi_exports_every_declared_symbol():
    import daqp_amd
    from daqp_amd._lib import EXPORTS
    L = daqp_amd.lib()
    hdr = open(os.path.join(ROOT, "include", "daqp_amd.h")).read()
    declared = set(re.findall(r"\b((?:daqp|setup_daqp|allocate_daqp|free_daqp)\w*)\s*\(", hdr))
    declared = {d for d in declared if not d.startswith("daqp_ldp")}
    assert declared, "no declarations parsed"
    for sym in declared | set(EXPORTS):
        assert hasattr(L, sym), f"libdaqp_amd.so does not export {sym}"
    assert b"daqp_amd" in L.daqp_amd_version()


def test_struct_layouts_match_reference_abi():
    """sizeof of the ctypes mirrors == the reference's x86-64 layouts"""
    from daqp_amd._lib import DAQPProblem, DAQPResult, DAQPSettings, WORKSPACE_BYTES
    assert C.sizeof(DAQPProblem) == 80 and C.sizeof(DAQPSettings) == 120 and C.sizeof(DAQPResult) == 64
    assert WORKSPACE_BYTES == 288
    assert DAQPSettings.cycle_tol.offset == 40 and DAQPSettings.fval_bound.offset == 48 and DAQPSettings.time_limit.offset == 112
    assert DAQPResult.exitflag.offset == 32 and DAQPResult.setup_time.offset == 56


def test_no_cpu_fallback_without_device():
    """without a HIP device the product path refuses loudly instead of computing on the host"""
    import daqp_amd
    L = daqp_amd.lib()
    if L.daqp_amd_device_count() > 0:
        return
    q = O.generate_qp(6, 12, 0, 2, rng=1)
    x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    assert flag == -8 and "no HIP device" in daqp_amd.last_error()
    try:
        daqp_amd.BatchModel(4, 6, 12, 0)
        assert False, "BatchModel must raise without a device"
    except RuntimeError as e:
        assert "no HIP device" in str(e)


def test_default_settings_match_reference_constants():
    import daqp_amd
    s = daqp_amd.default_settings()
    o = O.default_settings()
    for name, _ in s._fields_:
        assert getattr(s, name) == getattr(o, name), name


def test_warm_start_helpers_match_reference():
    """daqp_dual_init_active / daqp_primal_init_active (api.c:579-633) are host-only: they must set exactly the sense bits
    the reference's own functions set (compared against oracle/_ref when the reference library travelled here, and against
    the documented rule otherwise: |lam| > 1e-12 -> ACTIVE, sign -> LOWER; |slack| < 1e-9 -> ACTIVE(+LOWER))."""
    import daqp_amd
    from daqp_amd._lib import DAQPProblem, c_double_p, c_int_p
    L = daqp_amd.lib()
    ref = None
    if O.reference_available():
        ref = O.Reference().lib
        for fn in (ref.daqp_dual_init_active, ref.daqp_primal_init_active):
            fn.argtypes = [C.c_void_p, c_double_p]
            fn.restype = None
    rng = np.random.default_rng(3)
    for trial in range(20):
        n, m, ms = 6, 14, 3
        q = O.generate_qp(n, m, ms, 3, rng=[9, trial])
        x, lam, *_ = O.Oracle().quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        lam = lam + (rng.random(m) < 0.2) * 1e-13        # values below the 1e-12 threshold must stay inactive
        for which, vec in (("dual", lam), ("primal", x)):
            out = []
            for lib_ in (L, ref):
                if lib_ is None:
                    continue
                sense = np.zeros(m, np.int32)
                H, f, A, bu, bl = (np.ascontiguousarray(q[k], np.float64) for k in ("H", "f", "A", "bupper", "blower"))
                v = np.ascontiguousarray(vec, np.float64)
                qp = DAQPProblem(n, m, ms, H.ctypes.data_as(c_double_p), f.ctypes.data_as(c_double_p), A.ctypes.data_as(c_double_p),
                                 bu.ctypes.data_as(c_double_p), bl.ctypes.data_as(c_double_p), sense.ctypes.data_as(c_int_p), None, 0, 0)
                fn = getattr(lib_, f"daqp_{which}_init_active")
                fn(C.byref(qp) if lib_ is L else C.cast(C.byref(qp), C.c_void_p), v.ctypes.data_as(c_double_p))
                out.append(sense.copy())
            if ref is not None:
                assert np.array_equal(out[0], out[1]), (which, trial, out)
            if which == "dual":
                want = np.where(np.abs(lam) > 1e-12, O.ACTIVE + np.where(lam < 0, O.LOWER, 0), 0).astype(np.int32)
                assert np.array_equal(out[0], want)
            else:
                assert (out[0][np.abs(lam) > 1e-6] & O.ACTIVE).all()     # rows active at the optimum have zero slack


def test_first_violating_host_helper():
    """daqp_first_violating (api.c:562-574, host-only): first violated row, or m"""
    import daqp_amd
    L = daqp_amd.lib()
    L.daqp_first_violating.argtypes = [C.POINTER(C.c_double)] * 4 + [C.c_int] * 3 + [C.c_double]
    L.daqp_first_violating.restype = C.c_int
    n, m, ms = 3, 5, 2
    A = np.array([[1.0, 1, 0], [0, 1, 1], [1, 0, -1]])
    bu, bl = np.array([1.0, 1, 2, 2, 0.5]), -np.array([1.0, 1, 2, 2, 0.5])
    dp = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.POINTER(C.c_double))
    for x, want in (([0, 0, 0], 5), ([2, 0, 0], 0), ([0.5, -1.5, 0], 1), ([1, 1, 1.5], 3), ([1, 0.2, 0.2], 4)):
        xv = np.array(x, float)
        assert L.daqp_first_violating(dp(xv), dp(A), dp(bu), dp(bl), n, m, ms, 1e-9) == want, (x, want)


def test_kernel_resource_budgets():
    """The code generator's own per-kernel report of the last build (daqp_amd/_lib.py keeps it next to the objects): the hot
    kernels stay within the scratch and occupancy they were tuned at.  An innocent-looking edit that sends a register array to
    scratch does not fail any parity test -- it cost the generic setup kernel 2.5x in round 3 (47 -> 119 ms on config C4)."""
    import pytest
    from daqp_amd import _lib
    _lib.build()
    res = _lib.kernel_resources()
    if not res:
        pytest.skip("no resource report next to the objects (prebuilt library)")
    budgets = {   # kernel: (scratch bytes per lane at most, waves per SIMD at least)
        "k_setup<true, 4, false, false>": (128, 2), "k_setup<false, 4, false, false>": (128, 2), "k_setup<true, 8, false, false>": (128, 2),
        "k_setup<true, 4, true, false>": (0, 2), "k_setup<false, 4, true, false>": (0, 2),
        "k_setup<true, 4, false, true>": (0, 2), "k_setup<false, 4, false, true>": (0, 2),      # (PART: daqp_update_ldp's partial masks)
        "k_setup_fast<16, false, true>": (0, 4), "k_setup_fast<32, false, true>": (0, 2), "k_setup_fast<56, false, true>": (256, 2),
        "k_setup_fast<64, false, true>": (400, 2), "k_setup_fast<56, false, false>": (256, 2), "k_setup_fast<56, true, false>": (256, 2),
        "k_setup_tiny<4>": (64, 1), "k_setup_m": (0, 2), "k_fact_wg": (0, 2),
        "k_setup_blk<4, 56, true>": (0, 2), "k_setup_blk<4, 56, false>": (0, 2), "k_setup_blk<4, 64, false>": (0, 2),       # (the C2 setup with zero scratch)
        "k_setup_blk<3, 48, false>": (0, 2), "k_setup_blk<3, 40, true>": (0, 2), "k_setup_blk<2, 32, false>": (0, 2), "k_setup_blk<2, 24, true>": (0, 2),
        "k_ldp_reg<3, 25, true, 0>": (0, 1), "k_ldp_reg<3, 25, false, 0>": (0, 1), "k_ldp_reg<2, 32, true, 0>": (0, 1),
        "k_ldp_reg<1, 6, true, 0>": (48, 4), "k_ldp_reg<1, 6, false, 0>": (64, 4), "k_ldp_reg<1, 8, true, 0>": (80, 4), "k_ldp_reg<1, 8, false, 0>": (96, 4),   # (four waves per SIMD: measured faster with these few spilled registers than three without)
        "k_ldp_reg<1, 13, true, 0>": (32, 3), "k_ldp_reg<1, 13, false, 0>": (48, 3),   # (n <= 26: three waves per SIMD, twelve waves' LDS fit a CU)
        "k_ldp_reg<1, 16, true, 0>": (0, 2), "k_ldp_reg<2, 16, true, 0>": (16, 2),
        "k_ldp_reg<3, 8, true, 0>": (0, 2), "k_ldp_reg<3, 8, false, 0>": (0, 2), "k_ldp_reg<4, 8, true, 0>": (0, 2), "k_ldp_reg<4, 8, false, 0>": (32, 2), "k_ldp_reg<1, 25, true, 0>": (0, 2), "k_ldp_reg<1, 25, false, 0>": (0, 2),   # (few variables / many rows; n <= 50 with one row block)
        "k_ldp_reg<3, 25, true, 2>": (64, 2), "k_ldp_reg<2, 32, true, 1>": (64, 2), "k_ldp_reg<3, 25, true, 1>": (96, 2),     # (the C2 / C5 iteration with an fp32 image of M, two waves per SIMD, <= 256 registers)
        "k_ldp<1, false, 0, 0>": (0, 2), "k_ldp<2, false, 0, 0>": (0, 2), "k_ldp<4, true, 0, 0>": (128, 2),
        "k_ldp_wg<2, false, false>": (64, 2), "k_ldp_wg<4, false, false>": (512, 2), "k_ldp_wg<2, true, false>": (64, 2), "k_ldp_wg<4, true, false>": (512, 2),
        "k_ldp_wg<4, false, true>": (512, 2), "k_ldp_wg<2, false, true>": (64, 2),
        "k_ldp_reg<4, 32, true, 1>": (0, 1), "k_ldp_reg<8, 16, true, 1>": (0, 1), "k_ldp_reg<6, 25, true, 1>": (0, 1), "k_ldp_reg<5, 32, true, 1>": (0, 1),        # the image alone, one wave per SIMD: 256 - 320 image registers     # the tiered launch: two four-wave workgroups per CU = the same two waves per SIMD
        "k_update": (0, 8),
    }
    for name, (scratch, occ) in budgets.items():
        assert name in res, (name, sorted(res)[:8])
        r = res[name]
        assert r["scratch"] <= scratch and r["occupancy"] >= occ, (name, r)


def test_bench_gpus_flag_decides_the_ranks():
    """bench.py --gpus N: N > 1 without a launcher re-executes as N ranks under torch.distributed.run on the loopback interface;
    under a launcher the world size must be N; N == 1 stays in-process (tests/test_gpu_bench_contract.py runs all of it on the GPU)"""
    import pytest
    import bench
    assert bench.launch_plan(1, {}, ["--gpus", "1"]) is None
    assert bench.launch_plan(4, {"WORLD_SIZE": "4", "RANK": "2"}, []) is None
    cmd = bench.launch_plan(8, {}, ["--gpus", "8", "--config", "C3", "--strong"])
    i = cmd.index("torch.distributed.run")
    assert cmd[i + 1:i + 6] == ["--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1"] and cmd[i + 6] == "--master-port"
    assert 1024 < int(cmd[i + 7]) < 65536 and cmd[i + 8] == os.path.join(ROOT, "bench.py") and cmd[i + 9:] == ["--gpus", "8", "--config", "C3", "--strong"]
    for gpus, ws in ((8, "1"), (1, "2"), (2, "8")):
        with pytest.raises(SystemExit, match=f"--gpus {gpus} but the launcher started WORLD_SIZE={ws}"):
            bench.launch_plan(gpus, {"WORLD_SIZE": ws}, [])


def test_bench_quotes_counters_only_for_the_loaded_library(tmp_path, monkeypatch):
    """roofline numbers derived from rocprofv3 counters come from committed passes; bench.py quotes a summary only when its stamp
    (library version + hash of daqp_amd/csrc) is that of the library it runs, and the batch size is the one profiled"""
    import json
    import bench
    now = bench.library_stamp()
    assert len(now["csrc_sha16"]) == 16 and now["version"].startswith("daqp_amd")
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "library_stamp", lambda: now)
    assert bench.committed_counters("C2", 100000)["stale"]
    entry = {"batch": 100000, "traffic_bytes_per_launch": 1.0e10, "issue": {"attainable_ms": 9.0}, "binding": {"resource": "x"}}
    (prof / "r09a_pmc_summary.json").write_text(json.dumps({"_stamp": {"version": now["version"], "csrc_sha16": "0" * 16}, "C2": entry}))
    r = bench.committed_counters("C2", 100000)
    assert r["stale"] and "was taken with library" in r["why"]
    (prof / "r09b_pmc_summary.json").write_text(json.dumps({"_stamp": now, "C2": entry}))
    r = bench.committed_counters("C2", 100000)
    assert "stale" not in r and r["traffic"] == 1.0e10 and r["issue"]["attainable_ms"] == 9.0 and r["traffic_source"] == "profiles/r09b_pmc_summary.json"
    assert bench.committed_counters("C2", 4096)["stale"] and bench.committed_counters("C4", 10000)["stale"]


def test_committed_counters_belong_to_this_build():
    """bench.py quotes HBM traffic and the issue-side counters from the newest profiles/r*_pmc_summary.json -- only when that summary is
    stamped with the version and the csrc hash of the library that is loaded (a kernel edit without a counter refresh
    must not leave stale numbers in a driver-run record).  This test is the reminder: the committed summary covers every BASELINE
    configuration at its bench batch size and was taken with the sources as they are now."""
    import bench
    for cfg, c in bench.CONFIGS.items():
        got = bench.committed_counters(cfg, c["per_gpu"])
        assert not got.get("stale"), (cfg, got.get("why"))
        assert got["traffic"] > 0 and got["issue"]["attainable_ms"] > 0, cfg


def test_setup_records_pair_bytes_and_time_of_the_same_kernel():
    """C4's setup is three launches; the committed counter summary carries ONE record per kernel -- that kernel's HBM-side bytes
    (its own rows of the --pmc passes) next to that kernel's average duration (its own row of the kernel trace) -- and bench.py builds
    `roofline.setup` record by record from them: `traffic` and `avg_launch_ms` of a record cannot come from different kernels."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    assert files
    doc = json.load(open(files[-1]))
    recs = doc["C4"].get("setup_launches")
    assert recs and len(recs) == 3, "the newest counter summary carries C4's setup launch by launch"
    names = [r["kernel"] for r in recs]
    assert len(set(names)) == 3 and any("k_fact_wg" in k for k in names) and any("k_setup_m" in k for k in names) and any(k.startswith("k_setup<") for k in names)
    stats_file = files[-1].replace("_pmc_summary.json", "_kernel_stats.csv")
    import csv
    import re
    trace = {re.sub(r"^void daqp_amd::|\(.*$", "", r["Name"]): float(r["AverageNs"]) * 1e-6 for r in csv.DictReader(open(stats_file))}
    raw = json.load(open(files[-1].replace("_pmc_summary.json", "_pmc_raw_C4.json")))
    for r in recs:
        assert r["avg_ms_kernel_trace"] and abs(r["avg_ms_kernel_trace"] - trace[r["kernel"]]) < 1e-9, r["kernel"]          # the time: this kernel's trace row
        assert abs(r["hbm_read_bytes"] - raw[r["kernel"]]["FETCH_SIZE"]["mean"] * 2048) < 1.0, r["kernel"]                  # the bytes: this kernel's counter rows
        assert abs(r["hbm_written_bytes"] - raw[r["kernel"]]["WRITE_SIZE"]["mean"] * 1024) < 1.0, r["kernel"]
    # and bench.py turns each into one roofline record whose numbers are those
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'for st in prof["setup_launches"]' in src and '"kernel": st["kernel"], "bound": "hbm", "avg_launch_ms": ms_k' in src
