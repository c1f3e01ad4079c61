"""Singular-Hessian batches through the proximal outer loop: rate, outer iterations, parity vs the oracle.
usage: python tools/prox_rate.py [N] [n] [m] [rank]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daqp_amd
from oracle import oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
m = int(sys.argv[3]) if len(sys.argv) > 3 else 150
rank = int(sys.argv[4]) if len(sys.argv) > 4 else (4 * n) // 5
qs = [O.generate_singular_qp(n, m, 0, rank, rng=[91, k]) for k in range(N)]
b = {k: np.stack([q[k] for q in qs]) for k in ("H", "f", "A", "bupper", "blower", "sense")}
mdl = daqp_amd.BatchModel(N, n, m, 0)
for rep in range(3):
    t0 = time.time()
    mdl.setup(b["H"], b["f"], b["A"], b["bupper"], b["blower"], b["sense"], init_mask=64)
    t1 = time.time()
    r = mdl.solve()
    t2 = time.time()
    ks, kv = mdl.kernel_ms()
    print(f"rep {rep}: setup {1e3*(t1-t0):.1f} ms (kernels {ks:.1f}), solve {1e3*(t2-t1):.1f} ms (kernels {kv:.1f}) -> {N/(t2-t0):.0f} QPs/s host-inclusive")
info = mdl.prox_info()
print("prox problems", int((info["n_prox"] > 0).sum()), "outer mean/max", info["outer"].mean(), info["outer"].max(),
      "iter mean/max", r["iter"].mean(), r["iter"].max(), "flags", np.unique(r["exitflag"], return_counts=True))
ora = O.Oracle(fast=True)
K = min(N, 256)
t0 = time.time()
ref = [ora.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"]) for q in qs[:K]]
t1 = time.time()
same_it = sum(int(ref[k][4] == r["iter"][k] and ref[k][3] == r["exitflag"][k]) for k in range(K))
dx = max(np.abs(ref[k][0] - r["x"][k]).max() for k in range(K))
print(f"oracle (1 core, release flags): {K/(t1-t0):.0f} QPs/s; identical iter+flag {same_it}/{K}; max|dx| {dx:.2e}")
