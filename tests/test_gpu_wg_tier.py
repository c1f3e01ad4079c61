"""The tiered launch of the workgroup kernel (k_ldp_wg<4, false, true>: cold solves of the default arithmetic at two four-wave workgroups per
CU, rows < r0 of the inverse factor W = L^-1 in LDS, the rest in the problem's slot of the stored factor; wg_kernel.hip.h, wg_ldp.hip.h WROW)
against the oracle (reference: src/factorization.c:21-151, src/auxiliary.c:314-354): small batches forced through it
(DAQP_AMD_WG_TIER_MIN_BATCH=1), the tier boundary moved down (DAQP_AMD_WG_R0) so that most of every factor lives in the HBM tier, the stored
iterate read back by warm solves, and the problems it hands to the launch behind it (soft rows, degenerate factors)."""
import numpy as np
import pytest
from oracle import oracle as O

pytestmark = pytest.mark.gpu
XTOL = 1e-9


@pytest.fixture
def tier(monkeypatch):
    monkeypatch.setenv("DAQP_AMD_WG_TIER_MIN_BATCH", "1")
    monkeypatch.delenv("DAQP_AMD_EXACT", raising=False)
    return monkeypatch


def compare(g, ref):
    assert np.array_equal(g["exitflag"], ref[3]), (g["exitflag"], ref[3])
    assert np.array_equal(g["iter"], ref[4]), (g["iter"], ref[4])
    assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1]))
    assert np.abs(g["x"] - ref[0]).max() < XTOL


@pytest.mark.parametrize("r0", [None, 100, 64, 33])
def test_c4_through_the_tiers(oracle, gpu_lib, tier, r0):
    """config C4 (working sets of 130-160 rows): the shipped boundary (what half the LDS holds, ~104 rows), and lower ones -- at 33 four fifths
    of the factor are in the HBM tier and the removals' sweeps cross the boundary all the time"""
    import daqp_amd
    if r0:
        tier.setenv("DAQP_AMD_WG_R0", str(r0))
    n, m, ms, na, seed, _ = O.CONFIGS["C4"]
    q = O.generate_batch(12, n, m, ms, na, seed, start=40)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    compare(g, ref)


@pytest.mark.parametrize("shape", [(129, 200, 10, 30), (229, 400, 20, 60), (187, 371, 0, 92), (229, 420, 0, 205), (150, 512, 0, 70), (255, 300, 40, 90),
                                   (120, 300, 0, 40), (127, 400, 5, 45), (116, 640, 0, 50)])     # (the last three: two-chunk shapes whose factor does not fit half the LDS: k_ldp_wg<2, false, true>)
@pytest.mark.parametrize("r0", [None, 40])
def test_shapes_through_the_tiers(oracle, gpu_lib, tier, shape, r0):
    """four-chunk shapes: simple bounds, odd n, fewer / more row blocks than the four waves, working sets that pass 191 rows (where the inverse
    factor is given up: such a problem goes to the launch behind)"""
    import daqp_amd
    if r0:
        tier.setenv("DAQP_AMD_WG_R0", str(r0))
    n, m, ms, na = shape
    q = O.generate_batch(8 if na < 100 else 3, n, m, ms, na, 2900 + n + na)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    compare(g, ref)


def test_warm_solves_read_the_tiered_factor(oracle, gpu_lib, tier):
    """a tiered cold solve leaves L with its rows beyond r0 written in place; the warm steps behind it (UPDATE_v, then UPDATE_d) start from
    that stored factor in the launch that holds everything in LDS -- iteration counts and active sets as the oracle's models"""
    import daqp_amd
    tier.setenv("DAQP_AMD_WG_R0", "48")
    n, m, ms, na = 160, 420, 6, 70
    N = 6
    q = O.generate_batch(N, n, m, ms, na, 3100)
    bm = daqp_amd.BatchModel(N, n, m, ms)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
    g = bm.solve()
    mods = []
    for k in range(N):
        md = oracle.model(n, m, ms)
        md.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        r = md.solve()
        assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4]
        mods.append(md)
    f, bu, bl = q["f"].copy(), q["bupper"].copy(), q["blower"].copy()
    rng = np.random.default_rng(11)
    for t in range(4):
        if t % 2 == 0:
            f = f + 0.05 * rng.standard_normal(f.shape)
            bm.update(f=f)
        else:
            sh = 0.02 * rng.standard_normal(bu.shape)
            bu = bu + sh; bl = bl + sh
            bm.update(bupper=bu, blower=bl)
        g = bm.solve()
        for k, md in enumerate(mods):
            if t % 2 == 0:
                md.update(daqp_amd.UPDATE_v, f=f[k])
            else:
                md.update(daqp_amd.UPDATE_d, bupper=bu[k], blower=bl[k])
            r = md.solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (t, k, g["iter"][k], r[4])
            assert np.array_equal(np.sign(g["lam"][k]), np.sign(r[1])) and np.abs(g["x"][k] - r[0]).max() < XTOL
    bm.close()


def test_degenerate_families_through_the_tiers(oracle, gpu_lib, tier):
    """near-duplicate rows, (dependent) equalities, soft rows at n = 129 ... 200: singular pivots, pivoting, refinement, repair -- whatever leaves
    the inverse-factor representation beyond r0 rows, and every problem with soft rows, is solved by the launch behind the tiered one"""
    import daqp_amd
    tier.setenv("DAQP_AMD_WG_R0", "40")
    mism = []
    for trial in range(24):
        rng = np.random.default_rng([299, trial])
        eps = 10.0 ** rng.uniform(-13, -2)
        n = int(rng.integers(129, 201)); m = int(rng.integers(n + 20, 3 * n)); ms = int(rng.integers(0, n // 3))
        na = int(rng.integers(n // 4, n - 4))
        q = O.generate_nasty(n, m, ms, na, eps, rng, n_dup=int(rng.integers(0, 6)), n_eq=int(rng.integers(0, 4)),
                             n_soft=int(rng.integers(0, 3)) if trial % 2 else 0, dep_eq=bool(rng.integers(0, 2)))
        ns = int((q["sense"] & 8).astype(bool).sum())
        bm = daqp_amd.BatchModel(1, n, m, ms, ns_max=ns)
        bm.setup(q["H"][None], q["f"][None], q["A"][None], q["bupper"][None], q["blower"][None], q["sense"][None],
                 init_mask=daqp_amd.UPDATE_unconstrained)
        g = bm.solve()
        bm.close()
        r = oracle.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        flag, it = int(g["exitflag"][0]), int(g["iter"][0])
        ok = flag == r[3] and it == r[4]
        if ok and flag > 0:
            G = np.vstack([np.eye(n)[:ms], q["A"]])
            ok = np.abs(g["x"][0] - r[0]).max() < 1e-9 and np.abs(G.T @ (g["lam"][0] - r[1])).max() < 1e-7
        if not ok:
            mism.append((trial, n, m, ms, ns, flag, r[3], it, r[4]))
    assert not mism, mism[:10]


def test_tiered_and_untiered_launch_agree(gpu_lib, tier):
    """the same C4 draws with the tiered launch forced and with it switched off (DAQP_AMD_NO_WG_TIER=1: eight waves, the whole factor in LDS):
    iteration counts, exit flags and active sets identical, x to 1e-11 (the two sum W's products over different wave slices)"""
    import daqp_amd
    n, m, ms, na = 200, 600, 0, 80
    q = O.generate_batch(4, n, m, ms, na, 44)
    g1 = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    tier.setenv("DAQP_AMD_NO_WG_TIER", "1")
    g0 = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(g1["iter"], g0["iter"]) and np.array_equal(g1["exitflag"], g0["exitflag"])
    assert np.abs(g1["x"] - g0["x"]).max() < 1e-11 and np.array_equal(np.sign(g1["lam"]), np.sign(g0["lam"]))


@pytest.mark.parametrize("r0", [None, 33])
def test_iteration_limit_inside_the_tiered_launch_then_on(oracle, gpu_lib, tier, r0):
    """iter_limit = 150 stops a C4 solve in the middle (exit -4, the working set at ~100-130 rows: across the tier boundary): the W -> L conversion at
    the end of the launch crosses the tiers, and the SECOND daqp_solve continues from that stored factor -- in the launch that holds everything in
    LDS -- exactly where the reference's workspace continues (daqp.c:6-108: iterations restart at 1, the working set and L stay)"""
    import daqp_amd
    if r0:
        tier.setenv("DAQP_AMD_WG_R0", str(r0))
    n, m, ms, na, seed, _ = O.CONFIGS["C4"]
    N = 6
    q = O.generate_batch(N, n, m, ms, na, seed, start=80)
    bm = daqp_amd.BatchModel(N, n, m, ms, iter_limit=150)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
    mods = []
    for k in range(N):
        md = oracle.model(n, m, ms, settings=O.default_settings(iter_limit=150))
        md.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        mods.append(md)
    seen = set()
    for rnd in range(3):
        g = bm.solve()
        for k, md in enumerate(mods):
            r = md.solve()
            assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (rnd, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
            seen.add(int(r[3]))
            if r[3] > 0:
                assert np.array_equal(np.sign(g["lam"][k]), np.sign(r[1])) and np.abs(g["x"][k] - r[0]).max() < XTOL
    assert -4 in seen and 1 in seen
    bm.close()
