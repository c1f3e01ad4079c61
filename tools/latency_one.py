"""Latency of the single-problem drop-in entry points (a batch of one on the GPU): daqp_quadprog per call, and
setup_daqp once + daqp_solve / daqp_update_ldp per call.  usage: python tools/latency_one.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daqp_amd
from oracle import oracle as O

for cfg in ("C1", "C2"):
    n, m, ms, na, seed, _ = O.CONFIGS[cfg]
    q = O.generate_qp(n, m, ms, na, rng=[seed, 0])
    args = (q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    for _ in range(300):    # (the first few hundred calls of a process run slower: clocks, lazy loading)
        daqp_amd.solve(*args)
    t0 = time.perf_counter()
    K = 100
    for _ in range(K):
        x, fval, flag, info = daqp_amd.solve(*args)
    t1 = time.perf_counter()
    mdl = daqp_amd.Model()
    mdl.setup(*args)
    for _ in range(300):
        mdl.update(f=q["f"]); mdl.solve()
    t2 = time.perf_counter()
    for _ in range(K):
        mdl.update(f=q["f"]); mdl.solve()
    t3 = time.perf_counter()
    print(f"{cfg} (n={n}, m={m}): daqp_quadprog {1e3*(t1-t0)/K:.3f} ms per call (iterations {info['iterations']}); "
          f"update_ldp(UPDATE_v)+daqp_solve on a kept workspace {1e3*(t3-t2)/K:.3f} ms per call")
lp = O.generate_lp(20, 50, 0, [5, 0])
for _ in range(3):
    daqp_amd.solve(None, lp["f"], lp["A"], lp["bupper"], lp["blower"], lp["sense"])
t0 = time.perf_counter()
for _ in range(50):
    x, fval, flag, info = daqp_amd.solve(None, lp["f"], lp["A"], lp["bupper"], lp["blower"], lp["sense"])
t1 = time.perf_counter()
print(f"LP n=20 m=50: daqp_quadprog {1e3*(t1-t0)/50:.3f} ms per call (exit {flag}, iterations {info['iterations']}, outer {info['nodes']})")
