"""Parity campaign for the proximal outer loop (singular Hessians, forced shifts, LPs) over random shapes, against the
oracle, in exact mode (bit-identical x, lam, fval expected) -- every kernel family: register (n+1 <= 64), generic
(n > 63) and spilled (n around 200).
usage: python tools/prox_campaign.py [shapes] [per_shape] [--big]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DAQP_AMD_EXACT"] = "1"
import daqp_amd
from oracle import oracle as O

nshape = int(sys.argv[1]) if len(sys.argv) > 1 else 120
per = int(sys.argv[2]) if len(sys.argv) > 2 else 24
big = "--big" in sys.argv
ora = O.Oracle()
bits = lambda a: np.ascontiguousarray(a, np.float64).view(np.uint64)
tot = bad = 0
flags = {}
t0 = time.time()
shapes = []
for s in range(nshape):
    rng = np.random.default_rng([500, s])
    n = int(rng.integers(2, 90)); m = int(rng.integers(n + 1, min(3 * n + 3, 250))); ms = int(rng.integers(0, min(n, 8) + 1)) if s % 2 else 0
    shapes.append((n, m, ms))
if big:
    shapes += [(150, 300, 0), (200, 420, 10), (120, 260, 5), (160, 330, 0), (130, 270, 4)]   # kinds sing, diag, range, lp, forced (120 shapes before them)
for s, (n, m, ms) in enumerate(shapes):
    kind = ["sing", "diag", "range", "lp", "forced"][s % 5]
    kw = {}
    if kind == "range":
        kw = dict(eps_prox=1e-2, eta_prox=1e-8)
    if kind == "forced":
        kw = dict(eps_prox=10.0 ** np.random.default_rng(s).uniform(-4, -1))
    if s % 7 == 6:
        kw["iter_limit"] = 40
    N = per if n < 120 else 6
    qs = []
    for k in range(N):
        if kind == "lp":
            qs.append(O.generate_lp(n, m, ms, [501, s, k], unbounded=(k % 8 == 5)))
        elif kind == "forced" and k % 2:
            q = O.generate_qp(n, m, ms, max(1, min(n // 3, m - ms - 1)), rng=[502, s, k])
            qs.append({kk: q[kk] for kk in ("H", "f", "A", "bupper", "blower", "sense")})
        else:
            qs.append(O.generate_singular_qp(n, m, ms, rank=1 + (k * 5) % max(1, n - 1), rng=[503, s, k],
                                             kind="diag" if kind == "diag" else "dense", in_range=(kind == "range")))
    if s % 3 == 1:   # equalities (at most 4: below the reference's elimination threshold) and soft rows inside the proximal loop
        vr = np.random.default_rng([504, s])
        qs = [O.add_sense_variety(q, ms, int(vr.integers(0, min(4, n - 1, m - ms) + 1)), int(vr.integers(0, 3)) if m - ms > 6 else 0, [505, s, k])
              for k, q in enumerate(qs)]
    st = O.default_settings(**kw)
    ref = [ora.quadprog(q.get("H"), q["f"], q["A"], q["bupper"], q["blower"], q["sense"], settings=st) for q in qs]
    b = {k: np.stack([q[k] for q in qs]) for k in ("f", "A", "bupper", "blower", "sense")}
    H = None if kind == "lp" else np.stack([q["H"] for q in qs])
    r = daqp_amd.solve_batch(H, b["f"], b["A"], b["bupper"], b["blower"], b["sense"], ms=ms, **kw)
    for k in range(N):
        x, lam, fval, flag, it = ref[k]
        flags[flag] = flags.get(flag, 0) + 1
        ok = r["exitflag"][k] == flag and (flag == -5 or r["iter"][k] == it)
        if ok and flag > 0:
            ok = np.array_equal(bits(r["x"][k]), bits(x)) and np.array_equal(bits(r["lam"][k]), bits(lam)) and bits(r["fval"][k]) == bits(fval)
        tot += 1
        if not ok:
            bad += 1
            if bad <= 10:
                print(f"MISMATCH shape {s} (n={n} m={m} ms={ms}) {kind} {kw} qp {k}: gpu {r['exitflag'][k]}/{r['iter'][k]} oracle {flag}/{it} "
                      f"dx {np.abs(r['x'][k] - x).max():.2e}")
print(f"prox campaign: {len(shapes)} shapes, {tot} problems, exit flags {dict(sorted(flags.items()))}: {tot - bad} identical "
      f"(flag, iterations; bitwise x, lam, fval when solved), {bad} mismatches; {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
