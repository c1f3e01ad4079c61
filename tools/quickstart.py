"""The README quick start, runnable: python tools/quickstart.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, daqp_amd
from daqp_amd.synthetic import generate_batch_torch
x, fval, exitflag, info = daqp_amd.solve(np.eye(2), np.array([2.0, 2.0]), np.zeros((0, 2)), np.ones(2), -np.ones(2), np.zeros(2, np.int32))
print(x, fval, exitflag)
q = generate_batch_torch(10_000, 50, 150, 0, 20, seed=42)
bm = daqp_amd.BatchModel(10_000, 50, 150, ms=0)
bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=daqp_amd.UPDATE_unconstrained)
res = bm.solve(out="torch")
bm.update(f=q["f"] * 1.01)
res = bm.solve(out="torch")
print(res["exitflag"].unique(), res["iter"].float().mean())
rng = np.random.default_rng(0)
N, n, m = 64, 6, 14
f = rng.standard_normal((N, n)); A = rng.standard_normal((N, m, n)); x0 = rng.standard_normal((N, n))
s = np.einsum("qmn,qn->qm", A, x0)
lp = daqp_amd.solve_batch(None, f, A, s + 1.0, s - 1.0)
print(np.unique(lp["exitflag"], return_counts=True))
