"""Parity of random large shapes (n 65..229: generic setup kernel, streamed / spilled solve kernels) against the oracle, exact mode.
usage: python tools/large_shapes.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DAQP_AMD_EXACT"] = "1"
import daqp_amd
from oracle import oracle as O
ora = O.Oracle()
bits = lambda a: np.ascontiguousarray(a).view(np.uint64)
bad = tot = 0
t0 = time.time()
for s in range(40):
    rng = np.random.default_rng([800, s])
    n = int(rng.integers(65, 230)); m = int(rng.integers(n + 10, min(3 * n, 640))); ms = int(rng.integers(0, 20)) if s % 2 else 0
    na = int(rng.integers(5, min(n, m - ms) // 2))
    N = 6
    q = O.generate_batch(N, n, m, ms, na, 900 + s)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    r = ora.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    ok = np.array_equal(g["exitflag"], r[3]) and np.array_equal(g["iter"], r[4]) and np.array_equal(bits(g["x"]), bits(r[0])) and np.array_equal(bits(g["lam"]), bits(r[1]))
    tot += N
    if not ok:
        bad += 1
        print("MISMATCH", s, n, m, ms, na, g["exitflag"], r[3], g["iter"], r[4], np.abs(g["x"] - r[0]).max())
print(f"large shapes (n 65..229): 40 shapes, {tot} QPs, mismatching shapes {bad}; {time.time() - t0:.0f} s")
