"""k_ldp_reg<NB, NP, true, IMG != 0> (csrc/wave_ldp_reg.hip.h, reg_kernel.hip.h): the C2 / C5 iteration with an fp32 IMAGE of M in the registers
(two waves per SIMD).  The image only SCREENS the feasibility scan -- a verdict is taken from it only when it is provably the fp64 scan's, anything
else is decided by the fp64 pass over the blocked image --, a row that enters the working set is fetched in fp64, the active-row cache is tiered
(LDS + an L2-resident scratch), and a working set beyond the rows the kernel holds is handed to the full-register kernel behind it.  Bar: the default
mode's (north_star): exit flags, iteration counts and active sets identical to the reference algorithm, |x - x_ref|_inf < 1e-9.  Every test forces the
image kernel onto small batches (DAQP_AMD_IMG_MIN_BATCH=1; by default it serves batches of >= 1536 problems) and walks the switches that size its
LDS: DAQP_AMD_IMG_ROWS (rows held at all; beyond: hand-over), DAQP_AMD_IMG_CACHE (rows of them in LDS; beyond: scratch tier)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
XTOL = 1e-9


@pytest.fixture(autouse=True)
def image_kernel(monkeypatch):
    monkeypatch.delenv("DAQP_AMD_EXACT", raising=False)
    monkeypatch.setenv("DAQP_AMD_NO_RECHECK", "1")      # the kernel's own verdicts
    monkeypatch.setenv("DAQP_AMD_IMG_MIN_BATCH", "1")
    for k in ("DAQP_AMD_IMG_ROWS", "DAQP_AMD_IMG_CACHE", "DAQP_AMD_IMG_WAVES", "DAQP_AMD_NO_IMG32"):
        monkeypatch.delenv(k, raising=False)


def tier(monkeypatch, rows, cache, waves=None):
    if rows:
        monkeypatch.setenv("DAQP_AMD_IMG_ROWS", str(rows))
    if cache:
        monkeypatch.setenv("DAQP_AMD_IMG_CACHE", str(cache))
    if waves:
        monkeypatch.setenv("DAQP_AMD_IMG_WAVES", str(waves))


def same(g, ref):
    assert np.array_equal(g["exitflag"], ref[3])
    assert np.array_equal(g["iter"], ref[4])
    assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1]))
    assert np.abs(g["x"] - ref[0]).max() < XTOL


@pytest.mark.parametrize("rows,cache,waves", [(0, 0, 0), (0, 0, 5), (44, 6, 0), (30, 10, 0), (12, 3, 0), (2, 1, 0)])
def test_image_kernel_c2(oracle, gpu_lib, monkeypatch, rows, cache, waves):
    """C2 draws: the shipped carve-up (8 workgroups per CU), everything in LDS (5 per CU), most of every working set in the scratch tier, and
    caps that hand over a third / nearly all / all of the batch to the full-register kernel in mid-solve"""
    import daqp_amd
    tier(monkeypatch, rows, cache, waves)
    n, m, ms, na, seed, _ = O.CONFIGS["C2"]
    q = O.generate_batch(1024, n, m, ms, na, seed, start=700000)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    same(g, ref)
    monkeypatch.setenv("DAQP_AMD_NO_IMG32", "1")
    g0 = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    assert np.array_equal(g0["iter"], g["iter"]) and np.abs(g0["x"] - g["x"]).max() < 1e-12    # (the full-register kernel on the same problems)


@pytest.mark.parametrize("shape", [(17, 129, 0, 6), (33, 140, 0, 12), (41, 160, 0, 15), (50, 129, 0, 20), (49, 151, 0, 24), (50, 160, 0, 35), (26, 133, 0, 25),
                                   (51, 100, 0, 18), (56, 120, 0, 20), (63, 128, 0, 30), (60, 64, 0, 25), (63, 65, 0, 40), (55, 128, 0, 50),   # (these six: k_ldp_reg<2, 32, true, 1>)
                                   (40, 100, 0, 14), (50, 128, 0, 22), (33, 65, 0, 30),        # (two row blocks on the (3,25) image: the split block is empty)
                                   (50, 161, 0, 20), (45, 192, 0, 18), (20, 180, 0, 8),        # (k_ldp_reg<3, 25, true, 1>: three full row blocks)
                                   (50, 150, 10, 20), (40, 140, 40, 15), (33, 130, 5, 12), (60, 120, 8, 20), (63, 128, 63, 25), (50, 192, 12, 18)])   # (simple bounds: rows < ms are rows of R^-1, dots start at their column)
@pytest.mark.parametrize("cache", [0, 4])
def test_image_kernel_shapes(oracle, gpu_lib, monkeypatch, shape, cache):
    """every shape the (3, 25) image serves has three row blocks with at most 32 rows in the last one (129 <= m <= 160) and 17 <= n <= 50:
    odd n (the last pair's partner is padding), the fewest and the most rows of the split block, working sets up to n - 1 rows; the (2, 32)
    image (full row blocks) serves 51 <= n <= 63 with m <= 128: one and two row blocks, working sets beyond what it holds (hand-over)"""
    import daqp_amd
    tier(monkeypatch, 0, cache)
    n, m, ms, na = shape
    q = O.generate_batch(96, n, m, ms, na, 4100 + n + m)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    same(g, ref)


@pytest.mark.parametrize("rows,cache", [(0, 0), (40, 5)])
def test_image_kernel_warm_sequences(oracle, gpu_lib, monkeypatch, rows, cache):
    """UPDATE_v and UPDATE_d steps fused into the solve launch (the image is rounded from the fp64 rows that also form d = b s + M v), every step
    against the oracle's own update + solve on a kept workspace"""
    import daqp_amd
    tier(monkeypatch, rows, cache)
    n, m, ms, na, seed, _ = O.CONFIGS["C2"]
    S = 256
    q = O.generate_batch(S, n, m, ms, na, seed, start=710000)
    bm = daqp_amd.BatchModel(S, n, m, ms)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=0)
    bm.solve()
    mods = []
    for k in range(S):
        md = oracle.model(n, m, ms)
        md.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        md.solve()
        mods.append(md)
    rng = np.random.default_rng(11)
    f, bu, bl = q["f"].copy(), q["bupper"].copy(), q["blower"].copy()
    for t in range(6):
        if t % 3 != 2:
            f = f + 0.05 * rng.standard_normal(f.shape)
            bm.update(f=f)
        else:
            sh = 0.02 * rng.standard_normal(bu.shape)
            bu, bl = bu + sh, bl + sh
            bm.update(bupper=bu, blower=bl)
        g = bm.solve()
        rx, rlam, rfl, rit = np.zeros((S, n)), np.zeros((S, m)), np.zeros(S, np.int32), np.zeros(S, np.int32)
        for k, md in enumerate(mods):
            if t % 3 != 2:
                md.update(daqp_amd.UPDATE_v, f=f[k])
            else:
                md.update(daqp_amd.UPDATE_d, bupper=bu[k], blower=bl[k])
            r = md.solve()
            rx[k], rlam[k], rfl[k], rit[k] = r[0], r[1], r[3], r[4]
        same(g, (rx, rlam, None, rfl, rit))
    bm.close()


@pytest.mark.parametrize("family", ["3x25", "2x32", "bounds"])
@pytest.mark.parametrize("cache", [0, 5])
def test_image_kernel_degenerate_cases(oracle, gpu_lib, monkeypatch, cache, family):
    """near-duplicate rows (relative distance 1e-13 ... 1e-2), equalities (some dependent), soft rows on the image kernel's shapes: the pivot
    cascade, singular directions, the refinement step (which hands a problem over when rows sit in the scratch tier) and the refactor repair;
    exit flag and iterations identical, x to 1e-9, the multipliers' combined effect to 1e-7 (an ill-determined pair may share its multiplier
    differently between two arithmetics)"""
    import daqp_amd
    tier(monkeypatch, 0, cache)
    mism, marks = [], set()
    for trial in range(60):
        rng = np.random.default_rng([299, trial])
        eps = 10.0 ** rng.uniform(-13, -2)
        if family == "bounds":      # simple bounds among the rows (the Gram column's start columns, rows of R^-1 in the image)
            n = int(rng.integers(17, 51)); m = int(rng.integers(129, 161)); ms = int(rng.integers(1, n + 1))
        elif family == "3x25":
            n = int(rng.integers(17, 51)); m = int(rng.integers(129, 161)); ms = 0
        else:       # k_ldp_reg<2, 32, true, 1>: 51 <= n <= 63, at most two row blocks
            n = int(rng.integers(51, 61)); m = int(rng.integers(n + 4, 129)); ms = 0
        na = int(rng.integers(n // 4, n - 4))
        q = O.generate_nasty(n, m, ms, na, eps, rng, n_dup=int(rng.integers(0, 6)), n_eq=int(rng.integers(0, 4)),
                             n_soft=int(rng.integers(0, 4)), dep_eq=bool(rng.integers(0, 2)))
        ns = int((q["sense"] & 8).astype(bool).sum())
        if n + ns + 1 > 64:
            continue
        bm = daqp_amd.BatchModel(1, n, m, ms, ns_max=ns)
        bm.enable_trace(1 << 15)
        bm.setup(q["H"][None], q["f"][None], q["A"][None], q["bupper"][None], q["blower"][None], q["sense"][None], init_mask=daqp_amd.UPDATE_unconstrained)
        g = bm.solve()
        tr = bm.read_trace(marks=True)[0]
        marks |= {int(e) for e in tr if abs(int(e)) >= daqp_amd.api.TRACE_MARK}
        bm.close()
        md = oracle.model(n, m, ms, ns)         # (the same entry: setup with the unconstrained shortcut, no equality elimination)
        sflag = md.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"], init_mask=daqp_amd.UPDATE_unconstrained)
        r = md.solve() if sflag >= 0 else (None, None, 0.0, sflag, 0)
        flag, it = int(g["exitflag"][0]), int(g["iter"][0])
        ok = flag == r[3] and it == r[4]
        if ok and flag > 0:
            G = np.vstack([np.eye(n)[:ms], q["A"]])
            ok = np.abs(g["x"][0] - r[0]).max() < XTOL and np.abs(G.T @ (g["lam"][0] - r[1])).max() < 1e-7
        if not ok:
            mism.append((trial, n, m, ns, flag, r[3], it, r[4]))
    assert not mism, mism[:10]
    assert len(marks) >= 1, "no degenerate branch was taken: the generator parameters no longer reach them"


def test_image_kernel_limits_and_failures(oracle, gpu_lib, monkeypatch):
    """iteration limit (-4 with iter = the limit), an infeasible problem (-1 with the reference's iteration count: no second pass here), crossed
    bounds (-1 from the setup, nothing solved), next to ordinary problems of the same batch"""
    import daqp_amd
    n, m, ms, na, seed, _ = O.CONFIGS["C2"]
    q = O.generate_batch(64, n, m, ms, na, seed, start=720000)
    q["bupper"][3, 7], q["blower"][3, 7] = -1.0, 1.0                       # crossed
    q["blower"][5, :] = q["bupper"][5, :] - 1e-3                            # a box nobody fits in: infeasible
    q["bupper"][5, :10] = -50.0; q["blower"][5, :10] = -51.0
    st = O.default_settings()
    st.iter_limit = 30
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms, settings=st)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms, iter_limit=30)
    assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4])
    assert g["exitflag"][3] == -1 and (g["exitflag"] == -4).any() and (g["exitflag"] == 1).any()
    okp = g["exitflag"] == 1
    assert np.abs(g["x"][okp] - ref[0][okp]).max() < XTOL
    # settings.time_limit (daqp.c:95-103): the device clock is read every 32nd iteration; a budget of 100 ns stops every problem that needs more
    # than 32 iterations there with -7, the others are untouched
    q = O.generate_batch(64, n, m, ms, na, seed, start=721000)
    free = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    lim = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms, time_limit=1e-7)
    long_ = free["iter"] > 32
    assert long_.any() and (lim["exitflag"][long_] == -7).all() and (lim["iter"][long_] == 32).all()
    assert np.array_equal(lim["exitflag"][~long_], free["exitflag"][~long_]) and np.array_equal(lim["iter"][~long_], free["iter"][~long_])


@pytest.mark.parametrize("cache,shape", [(0, (50, 150, 0, 20)), (6, (50, 150, 0, 20)), (0, (50, 220, 0, 18)), (0, (60, 300, 6, 20))])
def test_image_kernel_shared_structure(oracle, gpu_lib, monkeypatch, cache, shape):
    """a shared-structure batch (ONE H and A, per-problem f and bounds: daqp_batch_setup_shared) at C2's shape: every problem's image is rounded
    from the same blocked M, the exact rows of every append come from it; cold solve, then fused warm updates, each step against the oracle.
    The last two shapes: the image-only kernels (4,32) and (5,32) on a shared image (their updates run as a launch of their own)"""
    import daqp_amd
    tier(monkeypatch, 0, cache)
    n, m, ms, na = shape
    N = 48
    q0 = O.generate_qp(n, m, ms, na, rng=[921, n])
    rng = np.random.default_rng([922, n])
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
            assert np.abs(g["x"][k] - r[0]).max() < XTOL * max(1.0, np.abs(r[0]).max(), np.abs(r[1]).max()), (t, k)
    bm.close()


# ---- the image ALONE (k_ldp_reg<4,32,true,1>): shapes with no full-register kernel (n <= 63, m <= 256 beyond the (3,25) / (2,32) registers) -------------
IMAGE_ONLY_SHAPES = [(50, 193, 0, 18), (50, 256, 0, 20), (40, 250, 12, 14), (63, 129, 0, 22), (63, 192, 0, 25), (56, 150, 8, 20), (51, 256, 51, 18),
                     (17, 200, 0, 6), (33, 230, 5, 12), (63, 256, 0, 30), (20, 256, 3, 7),
                     (30, 300, 0, 10), (32, 512, 4, 12), (24, 450, 0, 9),         # (8,16)
                     (50, 300, 0, 18), (45, 384, 10, 16), (33, 330, 0, 12),       # (6,25)
                     (63, 300, 0, 25), (56, 320, 7, 20), (51, 257, 0, 18),        # (5,32)
                     (64, 150, 0, 22), (64, 256, 5, 25), (64, 320, 0, 30)]        # n = 64: 65 rows only while a constraint is exchanged at a full vertex -- held rows 64, k_ldp behind


@pytest.mark.parametrize("shape", IMAGE_ONLY_SHAPES)
def test_image_only_shapes(oracle, gpu_lib, monkeypatch, shape):
    """the shapes that ran the one-wave kernel with M streamed in every scan up to round 6: their fp32 image in 256 registers, one wave per SIMD,
    every working-set row held (LDS + scratch tier), k_ldp behind it with mode | 4.  Exit flags, iteration counts and active sets as the oracle's,
    x to 1e-9; the same draws with the image kernel switched off (DAQP_AMD_NO_IMG_ONLY=1) agree as well"""
    import daqp_amd
    monkeypatch.delenv("DAQP_AMD_EXACT", raising=False)
    n, m, ms, na = shape
    q = O.generate_batch(24, n, m, ms, na, 5200 + n + m)
    ref = oracle.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    for off in (False, True):
        if off:
            monkeypatch.setenv("DAQP_AMD_NO_IMG_ONLY", "1")
        g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
        assert np.array_equal(g["exitflag"], ref[3]) and np.array_equal(g["iter"], ref[4]), (off, g["iter"], ref[4])
        assert np.array_equal(np.sign(g["lam"]), np.sign(ref[1])) and np.abs(g["x"] - ref[0]).max() < XTOL


def test_image_only_warm_sequence_and_limits(oracle, gpu_lib, monkeypatch):
    """n = 50, m = 200: cold solve, then UPDATE_v / UPDATE_d steps (the stand-alone update kernel, then the image kernel from the stored working set),
    and an iteration limit hit inside the image kernel followed by a second solve that continues"""
    import daqp_amd
    monkeypatch.delenv("DAQP_AMD_EXACT", raising=False)
    n, m, ms, na = 50, 200, 4, 18
    N = 16
    q = O.generate_batch(N, n, m, ms, na, 5300)
    for limit in (None, 12):
        st = dict(iter_limit=limit) if limit else {}
        bm = daqp_amd.BatchModel(N, n, m, ms, **st)
        bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
        mods = []
        for k in range(N):
            md = oracle.model(n, m, ms, settings=O.default_settings(**st) if st else None)
            md.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
            mods.append(md)
        f, bu, bl = q["f"].copy(), q["bupper"].copy(), q["blower"].copy()
        rng = np.random.default_rng(13)
        for t in range(5):
            if t in (2, 4):
                f = f + 0.05 * rng.standard_normal(f.shape)
                bm.update(f=f)
            elif t == 3:
                sh = 0.02 * rng.standard_normal(bu.shape)
                bu = bu + sh; bl = bl + sh
                bm.update(bupper=bu, blower=bl)
            g = bm.solve()
            for k, md in enumerate(mods):
                if t in (2, 4):
                    md.update(daqp_amd.UPDATE_v, f=f[k])
                elif t == 3:
                    md.update(daqp_amd.UPDATE_d, bupper=bu[k], blower=bl[k])
                r = md.solve()
                assert g["exitflag"][k] == r[3] and g["iter"][k] == r[4], (limit, t, k, g["exitflag"][k], r[3], g["iter"][k], r[4])
                if r[3] > 0:
                    assert np.array_equal(np.sign(g["lam"][k]), np.sign(r[1])) and np.abs(g["x"][k] - r[0]).max() < XTOL
        bm.close()


def test_image_only_degenerate_family(oracle, gpu_lib, monkeypatch):
    """near-duplicate rows, equalities (some dependent), soft rows on the image-only shapes: singular pivots, pivoting, refinement, repair"""
    import daqp_amd
    monkeypatch.delenv("DAQP_AMD_EXACT", raising=False)
    mism = []
    for trial in range(60):
        rng = np.random.default_rng([399, trial])
        eps = 10.0 ** rng.uniform(-13, -2)
        if trial % 4 == 1:
            n = int(rng.integers(51, 63)); m = int(rng.integers(129, 321))       # (4,32) / (5,32)
        elif trial % 4 == 3:
            n = int(rng.integers(18, 33)); m = int(rng.integers(257, 513))       # (8,16)
        else:
            n = int(rng.integers(18, 51)); m = int(rng.integers(193, 385))       # (4,32) / (6,25)
        ms = int(rng.integers(0, n // 3)); na = int(rng.integers(max(2, n // 4), n - 4))
        q = O.generate_nasty(n, m, ms, na, eps, rng, n_dup=int(rng.integers(0, 6)), n_eq=int(rng.integers(0, 4)),
                             n_soft=int(rng.integers(0, 3)) if n < 55 else 0, dep_eq=bool(rng.integers(0, 2)))
        ns = int((q["sense"] & 8).astype(bool).sum())
        bm = daqp_amd.BatchModel(1, n, m, ms, ns_max=ns)
        bm.setup(q["H"][None], q["f"][None], q["A"][None], q["bupper"][None], q["blower"][None], q["sense"][None], init_mask=daqp_amd.UPDATE_unconstrained)
        g = bm.solve()
        bm.close()
        md = oracle.model(n, m, ms, ns)         # (the same entry: setup with the unconstrained shortcut, no equality elimination)
        sflag = md.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"], init_mask=daqp_amd.UPDATE_unconstrained)
        r = md.solve() if sflag >= 0 else (None, None, 0.0, sflag, 0)
        flag, it = int(g["exitflag"][0]), int(g["iter"][0])
        ok = flag == r[3] and it == r[4]
        if ok and flag > 0:
            G = np.vstack([np.eye(n)[:ms], q["A"]])
            ok = np.abs(g["x"][0] - r[0]).max() < XTOL and np.abs(G.T @ (g["lam"][0] - r[1])).max() < 1e-7
        if not ok:
            mism.append((trial, n, m, ms, ns, flag, r[3], it, r[4]))
    assert not mism, mism[:10]
