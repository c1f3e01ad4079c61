"""Parity of random large shapes (n 65..229: generic setup kernel, workgroup solve kernel) against the oracle: exact mode (bit for
bit), then the default mode (exit flags, iterations, active sets identical, |dx| < 1e-9), then 256 QPs of config C4 in both.
usage: python tools/large_shapes.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daqp_amd
from oracle import oracle as O
ora = O.Oracle()
bits = lambda a: np.ascontiguousarray(a).view(np.uint64)
def campaign(exact):
    os.environ["DAQP_AMD_EXACT"] = "1" if exact else "0"
    bad = tot = 0
    dxm = 0.0
    t0 = time.time()
    cases = [(s, None) for s in range(40)] + [("C4", 256)]
    for s, NC in cases:
        if s == "C4":
            n, m, ms, na, seed, _ = O.CONFIGS["C4"]
            N = NC
            q = O.generate_batch(N, n, m, ms, na, seed)
        else:
            rng = np.random.default_rng([800, s])
            n = int(rng.integers(65, 230)); m = int(rng.integers(n + 10, min(3 * n, 640))); ms = int(rng.integers(0, 20)) if s % 2 else 0
            na = int(rng.integers(5, min(n, m - ms) // 2))
            N = 6
            q = O.generate_batch(N, n, m, ms, na, 900 + s)
        try:
            g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
        except RuntimeError as e:
            print("FAILED at case", s, (n, m, ms, na), "N", N, e, flush=True)
            raise
        r = ora.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
        ok = np.array_equal(g["exitflag"], r[3]) and np.array_equal(g["iter"], r[4])
        if exact:
            ok = ok and np.array_equal(bits(g["x"]), bits(r[0])) and np.array_equal(bits(g["lam"]), bits(r[1]))
        else:
            ok = ok and np.array_equal(np.sign(g["lam"]), np.sign(r[1])) and np.abs(g["x"] - r[0]).max() < 1e-9
        dxm = max(dxm, float(np.abs(g["x"] - r[0]).max()))
        tot += N
        if not ok:
            bad += 1
            print("MISMATCH", s, n, m, ms, na, g["exitflag"], r[3], g["iter"], r[4], np.abs(g["x"] - r[0]).max())
    print(f"{'exact' if exact else 'default'} mode: {len(cases)} shapes (n 65..229 + C4), {tot} QPs, mismatching shapes {bad}, max|dx| {dxm:.2e}, {time.time() - t0:.1f} s", flush=True)
    return bad


sys.exit(1 if (campaign(True) + campaign(False)) else 0)
