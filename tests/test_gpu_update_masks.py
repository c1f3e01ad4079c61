"""Every daqp_update_ldp mask of the reference (utils.c:58-221; its own binding builds them field by field, daqp.pyx:513-571)
through the HIP path: tests/golden/golden_update_masks.npz -- written by the reference library (tests/golden/make_golden_masks.py) --
replayed through Model (the single-problem drop-in symbols), BatchModel (daqp_batch_update), MultiBatchModel (daqp_batch_update_multi)
and a gcc-compiled C caller.

exact mode (DAQP_AMD_EXACT=1): update flags, x, lam, fval, iteration counts, exit flags and working sets bit for bit.
default mode: update flags, exit flags, iteration counts, working sets and active sets identical, |x - x_ref| < 1e-9.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import mask_replay as MR

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, M, V, D, S = 1, 2, 4, 8, 16
FULL = R | M | V | D | S


def _ws_of(model):
    """working set of a single-problem workspace: work->WS[0 .. n_active) (host mirrors, types.h:187-264)"""
    na = C.c_int.from_buffer(model._ws, 184).value
    p = C.cast(C.c_void_p.from_buffer(model._ws, 176).value, C.POINTER(C.c_int))
    return np.array([p[i] for i in range(na)], np.int32)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("shape", list(MR.SHAPES))
def test_update_masks_single_problem(gpu_lib, monkeypatch, shape, exact):
    """daqp_update_ldp(mask) + daqp_solve on a kept workspace, mask by mask, against the reference's own sequences"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    n, m, ms = MR.SHAPES[shape]
    checked = 0
    for mask, steps in MR.sequences(shape):
        for trial in range(MR.TRIALS):
            q = MR.base(shape, trial)
            d = daqp_amd.Model()
            flag, _ = d.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
            assert flag == 1
            seq = MR.Sequence(exact)
            for s in range(-1, steps):
                kw, exp = MR.step(shape, trial, mask, s)
                uflag = d.update_mask(mask, **kw) if s >= 0 else 0
                x, fval, ef, info = d.solve()
                seq.step(f"{shape}/{trial}/mask {mask}/step {s}", dict(x=x, lam=info["lam"], fval=fval, flag=ef, iter=info["iterations"],
                                                                   uflag=uflag, ws=_ws_of(d)), exp)
                checked += 1
    assert checked >= 40


def test_model_update_builds_the_mask_field_by_field(gpu_lib, monkeypatch):
    """Model.update(A=...) is daqp_update_ldp(UPDATE_M), not a re-setup (daqp.pyx:513-571): the bits follow the arrays given, and
    arrays of the wrong shape are ignored as the reference ignores them"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms = MR.SHAPES["c3"]
    for mask in (M, R, S, M | D, R | S, V | D | S):
        q = MR.base("c3", 1)
        d = daqp_amd.Model()
        assert d.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])[0] == 1
        d.solve()
        for s in range(3):
            kw, exp = MR.step("c3", 1, mask, s)
            args = dict(kw)
            if not (mask & V):
                args["f"] = np.zeros(3)        # (a 3-vector is no f of this problem: ignored, the v bit stays clear)
            assert d.update(**args) == exp["uflag"]
            x, fval, ef, info = d.solve()
            MR.check(f"c3/1/mask {mask}/step {s}", dict(x=x, lam=info["lam"], fval=fval, flag=ef, iter=info["iterations"], uflag=exp["uflag"]), exp, True)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("shape", list(MR.SHAPES))
def test_update_masks_batch(gpu_lib, monkeypatch, shape, exact):
    """daqp_batch_update(mask): the fixture's trials of one (shape, mask) as ONE batch -- every problem its own arrays"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    n, m, ms = MR.SHAPES[shape]
    N = MR.TRIALS
    for mask, steps in MR.sequences(shape):
        qs = [MR.base(shape, t) for t in range(N)]
        bm = daqp_amd.BatchModel(N, n, m, ms, MR.NS[shape])
        bm.setup(*(np.stack([q[k] for q in qs]) for k in ("H", "f", "A", "bupper", "blower", "sense")))
        cur = [dict(q) for q in qs]
        failed = np.zeros(N, bool)
        seqs = [MR.Sequence(exact) for _ in range(N)]
        for s in range(-1, steps):
            exps = []
            for t in range(N):
                kw, exp = MR.step(shape, t, mask, s)
                cur[t].update(kw)
                exps.append(exp)
            if s >= 0:
                names = [k for k in MR.ARRAYS if k in MR.step(shape, 0, mask, s)[0]]
                bm.update(mask=mask, **{k: np.stack([c[k] for c in cur]) for k in names})
                fl = bm.setup_flags()
                for t in range(N):
                    assert (fl[t] if fl[t] < 0 else 0) == exps[t]["uflag"], (shape, mask, s, t, fl[t], exps[t]["uflag"])
                failed = fl < 0
            g = bm.solve()
            na, ws = bm.working_sets()
            for t in range(N):
                if failed[t]:      # the batch API reports a failed update from the next solve (include/daqp_amd.h); the drop-in symbol solves on
                    assert g["exitflag"][t] == exps[t]["uflag"] and g["iter"][t] == 0
                    continue
                seqs[t].step(f"{shape}/{t}/mask {mask}/step {s}", dict(x=g["x"][t], lam=g["lam"][t], fval=g["fval"][t], flag=int(g["exitflag"][t]),
                                                                  iter=int(g["iter"][t]), uflag=exps[t]["uflag"], ws=ws[t, : na[t]]), exps[t])
        bm.close()


def test_update_masks_after_a_failed_batch_update_the_state_is_the_references(gpu_lib, monkeypatch):
    """a failed daqp_batch_update (crossed bounds) leaves the LDP as it was: the step after the repair matches the reference's
    sequence, which solved on the old LDP in between (fixture: shape mix, trial 1)"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    n, m, ms = MR.SHAPES["mix"]
    for mask in (D, V | D, M | D, R | D, D | S):
        q = MR.base("mix", 1)
        bm = daqp_amd.BatchModel(1, n, m, ms, MR.NS["mix"])
        bm.setup(*(q[k][None] for k in ("H", "f", "A", "bupper", "blower", "sense")))
        bm.solve()
        kw, exp = MR.step("mix", 1, mask, 0)
        assert exp["uflag"] == -1
        bm.update(mask=mask, **{k: v[None] for k, v in kw.items()})
        assert bm.setup_flags()[0] == -1
        kw, exp = MR.step("mix", 1, mask, 1)
        bm.update(mask=mask, **{k: v[None] for k, v in kw.items()})
        assert bm.setup_flags()[0] == 1
        g = bm.solve()
        # (iterations may differ from the fixture's: the reference's intermediate solve moved its working set; the optimum may not)
        assert g["exitflag"][0] == exp["flag"]
        if exp["flag"] > 0:
            assert np.abs(g["x"][0] - exp["x"]).max() < 1e-9 and np.array_equal(np.sign(g["lam"][0]), np.sign(exp["lam"]))
        bm.close()


@pytest.mark.parametrize("exact", [True, False])
def test_sense_bit_without_a_sense_array(oracle, gpu_lib, monkeypatch, exact):
    """DAQP_UPDATE_sense with qp->sense == NULL (utils.c:85-86): the workspace's sense becomes all zeros and NOTHING is rebuilt -- the
    rows of the working set lose their ACTIVE / LOWER bits while they stay in it.  The reference iterates on from that state; so does
    this path (in the default mode such a workspace runs the exact kernels until its next update: the default kernels' carried
    intermediate results assume the flags they were formed with)"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1" if exact else "0")
    for shape in ("c1", "c3", "mix"):
        n, m, ms = MR.SHAPES[shape]
        N = MR.TRIALS
        qs = [MR.base(shape, t) for t in range(N)]
        bm = daqp_amd.BatchModel(N, n, m, ms, MR.NS[shape])
        bm.setup(*(np.stack([q[k] for q in qs]) for k in ("H", "f", "A", "bupper", "blower", "sense")))
        oms = []
        for q in qs:
            om = oracle.model(n, m, ms, ns=MR.NS[shape])
            assert om.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"]) == 1
            oms.append(om)
        seqs = [MR.Sequence(exact) for _ in range(N)]
        for step, mask in enumerate((0, S, S | D, 0, V)):
            rng = np.random.default_rng([5, step])
            bu = np.stack([q["bupper"] for q in qs]) + 0.01 * rng.random((N, m))
            bl = np.stack([q["blower"] for q in qs])
            if shape == "mix":
                bl[:, ms + 1] = bu[:, ms + 1]
            f = np.stack([q["f"] for q in qs]) + 0.1 * rng.standard_normal((N, n))
            if mask:
                bm.update(mask=mask, **(dict(bupper=bu, blower=bl) if mask & D else {}), **(dict(f=f) if mask & V else {}))
                for t in range(N):
                    assert oms[t].o.lib.ora_update(oms[t].h, mask, None, O_dp(f[t]) if mask & V else None, None,
                                                   O_dp(bu[t]) if mask & D else None, O_dp(bl[t]) if mask & D else None, None) == 0
            g = bm.solve()
            for t in range(N):
                r = oms[t].solve()
                seqs[t].step(f"{shape}/{t}/step {step} mask {mask}", dict(x=g["x"][t], lam=g["lam"][t], fval=g["fval"][t], flag=int(g["exitflag"][t]),
                                                                      iter=int(g["iter"][t]), uflag=0),
                             dict(x=r[0], lam=r[1], fval=r[2], flag=r[3], iter=r[4], uflag=0))
        bm.close()


def O_dp(a):
    a = np.ascontiguousarray(a, np.float64)
    O_dp.keep.append(a)
    return a.ctypes.data_as(C.POINTER(C.c_double))


O_dp.keep = []


def test_update_masks_multi_device_entry(gpu_lib, monkeypatch):
    """daqp_batch_update_multi with partial masks: one device listed twice, against the fixture"""
    import daqp_amd
    monkeypatch.setenv("DAQP_AMD_EXACT", "1")
    shape = "c3"
    n, m, ms = MR.SHAPES[shape]
    N = MR.TRIALS
    for mask in (M, R | D, S, M | V | D | S):
        steps = dict(MR.sequences(shape))[mask]
        qs = [MR.base(shape, t) for t in range(N)]
        mb = daqp_amd.MultiBatchModel(N, n, m, ms, 0, devices=[0, 0])
        mb.setup(*(np.stack([q[k] for q in qs]) for k in ("H", "f", "A", "bupper", "blower", "sense")))
        cur = [dict(q) for q in qs]
        for s in range(-1, steps):
            exps = []
            for t in range(N):
                kw, exp = MR.step(shape, t, mask, s)
                cur[t].update(kw)
                exps.append(exp)
            if s >= 0:
                names = [k for k in MR.ARRAYS if k in MR.step(shape, 0, mask, s)[0]]
                mb.update(mask=mask, **{k: np.stack([c[k] for c in cur]) for k in names})
            g = mb.solve()
            for t in range(N):
                MR.check(f"multi {shape}/{t}/mask {mask}/step {s}", dict(x=g["x"][t], lam=g["lam"][t], fval=g["fval"][t], flag=int(g["exitflag"][t]),
                                                                    iter=int(g["iter"][t]), uflag=exps[t]["uflag"]), exps[t], True)
        mb.close()


def test_no_accepted_mask_is_unsupported(gpu_lib):
    """all 32 combinations of the five LDP bits are taken by daqp_batch_update and by daqp_update_ldp (none returns -8)"""
    import daqp_amd
    n, m, ms = MR.SHAPES["c3"]
    q = MR.base("c3", 0)
    bm = daqp_amd.BatchModel(2, n, m, ms)
    st = lambda k: np.stack([q[k], q[k]])
    bm.setup(st("H"), st("f"), st("A"), st("bupper"), st("blower"), st("sense"))
    d = daqp_amd.Model()
    assert d.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])[0] == 1
    for mask in range(32):
        bm.update(mask=mask, H=st("H"), f=st("f"), A=st("A"), bupper=st("bupper"), blower=st("blower"), sense=st("sense"))
        assert (bm.setup_flags() == 1).all(), mask
        g = bm.solve()
        assert (g["exitflag"] == 1).all(), mask
        assert d.update_mask(mask) == 0, mask
        assert d.solve()[2] == 1, mask
        # the unconstrained bit on top (daqp_quadprog's setup mask, api.c:62-79) is taken as well
        assert d.update_mask(mask | 64) == 0, mask
        assert d.solve()[2] == 1, mask
    bm.close()


def _build_mask_caller(tmp_path):
    import daqp_amd
    daqp_amd.lib()
    exe = os.path.join(str(tmp_path), "mask_caller")
    libdir, rocm = os.path.join(ROOT, "daqp_amd", "lib"), "/opt/rocm/lib"
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "mask_caller.c"), "-L" + libdir, "-ldaqp_amd", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath-link," + rocm, "-Wl,-rpath," + rocm, "-o", exe])
    return exe


def _write_sequence(path, shape, trial, mask, steps):
    import struct
    n, m, ms = MR.SHAPES[shape]
    q = MR.base(shape, trial)
    dts = dict(H=np.float64, f=np.float64, A=np.float64, bupper=np.float64, blower=np.float64, sense=np.int32)
    with open(path, "wb") as fp:
        fp.write(struct.pack("4i", n, m, ms, steps))
        for k in MR.ARRAYS:
            fp.write(np.ascontiguousarray(q[k], dts[k]).tobytes())
        for s in range(steps):
            kw, _ = MR.step(shape, trial, mask, s)
            fp.write(struct.pack("i", mask))
            for k in MR.ARRAYS:
                fp.write(struct.pack("i", 1 if k in kw else 0))
                if k in kw:
                    fp.write(np.ascontiguousarray(kw[k], dts[k]).tobytes())


def _parse_sequence(out):
    vec = lambda ln: np.array([float.fromhex(t) for t in ln.split()[1:]])
    res, upd = [], [0]
    lines = out.strip().splitlines()
    i = 0
    while i < len(lines):
        t = lines[i].split()
        if t[0] == "update":
            upd.append(int(t[1]))
        elif t[0] == "solve":
            res.append(dict(flag=int(t[3]), iter=int(t[5]), fval=float.fromhex(t[9]), uflag=upd[-1],
                            ws=np.array([int(v) for v in lines[i + 1].split()[1:]], np.int32), x=vec(lines[i + 2]), lam=vec(lines[i + 3])))
            i += 3
        i += 1
    return res


@pytest.mark.parametrize("exact", ["1", "0"])
def test_update_masks_from_compiled_c(tmp_path, gpu_lib, exact):
    """tests/c/mask_caller.c (gcc, include/daqp_amd.h, -ldaqp_amd): the reference's C call sequence with partial masks"""
    exe = _build_mask_caller(tmp_path)
    env = dict(os.environ, DAQP_AMD_EXACT=exact)
    seq = os.path.join(str(tmp_path), "seq.bin")
    for shape, trial, mask in (("c1", 0, M), ("c1", 1, R), ("c3", 2, M | D), ("c3", 0, S), ("mix", 1, R | D), ("mix", 1, M | V | D | S),
                               ("mix", 2, R | S), ("wide", 0, M), ("c3", 1, R | M | D | S)):
        steps = dict(MR.sequences(shape))[mask]
        _write_sequence(seq, shape, trial, mask, steps)
        r = subprocess.run([exe, seq], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (shape, trial, mask, r.returncode, r.stderr[-400:], r.stdout[-300:])
        got = _parse_sequence(r.stdout)
        assert len(got) == steps + 1, r.stdout[:400]
        seqc = MR.Sequence(exact == "1")
        for s in range(-1, steps):
            seqc.step(f"C {shape}/{trial}/mask {mask}/step {s}", got[s + 1], MR.step(shape, trial, mask, s)[1])
