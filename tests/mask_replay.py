"""Replay of tests/golden/golden_update_masks.npz (written by the reference library: tests/golden/make_golden_masks.py):
setup_daqp -> daqp_solve -> {daqp_update_ldp(mask, arrays) -> daqp_solve}* for every mask of utils.c:58-221.
Shared by the CPU test of the oracle and the GPU tests of the HIP path."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_update_masks.npz")
SHAPES = {"c1": (20, 40, 0), "c3": (12, 48, 12), "mix": (10, 30, 4), "wide": (66, 100, 4)}
NS = {"c1": 0, "c3": 0, "mix": 3, "wide": 0}     # soft rows of the shape's base sense (capacity of the working set)
TRIALS = 3
ARRAYS = ("H", "f", "A", "bupper", "blower", "sense")
_cache = {}


def golden():
    if "g" not in _cache:
        z = np.load(GOLDEN)
        _cache["g"] = {k: z[k] for k in z.files}
    return _cache["g"]


def sequences(shape):
    """[(mask, steps)] of a shape, steps = how many update + solve pairs the fixture holds"""
    g = golden()
    out = []
    for mask in g["masks"]:
        steps = 0
        while f"{shape}/0/{int(mask)}/{steps}/res" in g:
            steps += 1
        if steps:
            out.append((int(mask), steps))
    return out


def base(shape, trial):
    g = golden()
    return {k: g[f"{shape}/{trial}/{k}"] for k in ARRAYS}


def step(shape, trial, mask, s):
    """(arrays handed over, expected dict) of step s; s = -1: the solve after the setup"""
    g = golden()
    pre = f"{shape}/{trial}/{mask}"
    if s < 0:
        r = g[f"{pre}/res0"]
        return {}, dict(x=g[f"{pre}/x0"], lam=g[f"{pre}/lam0"], fval=r[0], flag=int(r[1]), iter=int(r[2]), uflag=0, ws=g[f"{pre}/ws0"])
    sp = f"{pre}/{s}"
    kw = {k: g[f"{sp}/{k}"] for k in ARRAYS if f"{sp}/{k}" in g}
    r = g[f"{sp}/res"]
    return kw, dict(x=g[f"{sp}/x"], lam=g[f"{sp}/lam"], fval=r[0], flag=int(r[1]), iter=int(r[2]), uflag=int(r[3]), ws=g[f"{sp}/ws"])


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def check(tag, got, exp, exact, tol=1e-9, tainted=False, warm=True):
    """got: dict(x, lam, fval, flag, iter, uflag[, ws]) of the path under test; exp: the reference's.
    exact: bitwise (the reference's arithmetic); otherwise the north_star bar: exit flag, iteration count, working set, active set
    with sides, |x - x_ref| < tol.
    Returns True when this step taints the rest of its sequence in the default arithmetic: an INFEASIBLE verdict of a WARM solve.
    That verdict falls in the singular branch, where components of a singular direction that are zero in exact arithmetic are compared
    with dual_tol (auxiliary.c:284-287, daqp.c:86-93): rounding noise decides whether one more row is removed before the certificate,
    and only the reference's arithmetic has the reference's noise (a first solve after a setup is re-derived in that arithmetic,
    recheck.hip.h; a warm solve starts from a state that already carries the default mode's rounding).  The exit flag is the same;
    the iteration count and the working set left behind may differ by that row -- and a later update that keeps the workspace's
    sense (utils.c:84-91) then starts from other stale ACTIVE bits.  `tainted`: compare what the north_star asks for from here on --
    exit flag, and for a solved problem the active set and x -- not the path."""
    assert got["uflag"] == exp["uflag"], f"{tag}: update flag {got['uflag']} != {exp['uflag']}"
    assert got["flag"] == exp["flag"], f"{tag}: exit flag {got['flag']} != {exp['flag']}"
    if exact:
        tainted = False
    warm_infeasible = (not exact) and warm and exp["flag"] == -1
    if warm_infeasible:
        if not tainted:
            assert abs(got["iter"] - exp["iter"]) <= 2, f"{tag}: iterations {got['iter']} vs {exp['iter']} on an infeasible problem"
        return True
    if not tainted:
        assert got["iter"] == exp["iter"], f"{tag}: iterations {got['iter']} != {exp['iter']}"
        if got.get("ws") is not None:
            assert np.array_equal(np.asarray(got["ws"]), exp["ws"]), f"{tag}: working set {got['ws']} != {exp['ws']}"
    if exp["flag"] < 0:
        return tainted
    if exact:
        assert np.array_equal(bits(got["x"]), bits(exp["x"])), f"{tag}: x differs ({np.abs(got['x'] - exp['x']).max():.2e})"
        assert np.array_equal(bits(got["lam"]), bits(exp["lam"])), f"{tag}: lam differs"
        assert bits(got["fval"]) == bits(exp["fval"]), f"{tag}: fval differs"
    else:
        assert np.array_equal(np.sign(got["lam"]), np.sign(exp["lam"])), f"{tag}: active set differs"
        assert np.abs(got["x"] - exp["x"]).max() < tol, f"{tag}: |x - x_ref| = {np.abs(got['x'] - exp['x']).max():.2e}"
        assert np.abs(got["lam"] - exp["lam"]).max() < 1e-6 * (1 + np.abs(exp["lam"]).max())
        assert abs(got["fval"] - exp["fval"]) < 1e-8 * (1 + abs(exp["fval"]))
    return tainted


class Sequence:
    """the steps of one workspace, with the taint of `check` carried along"""

    def __init__(self, exact):
        self.exact, self.tainted, self.first = exact, False, True

    def step(self, tag, got, exp):
        self.tainted = check(tag, got, exp, self.exact, tainted=self.tainted, warm=not self.first)
        self.first = False
