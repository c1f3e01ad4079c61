"""Batches that go through the proximal outer loop (prox.hip.h): rate, outer iterations, parity with the oracle.
usage: python tools/prox_rate.py [kind] [N] [n] [m]     kind: sing (rank-deficient H, vertex minimiser),
       range (rank-deficient H, f in range(H): many outer iterations, eps_prox=1e-2, eta_prox=1e-8), lp (H = None)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daqp_amd
from oracle import oracle as O

kind = sys.argv[1] if len(sys.argv) > 1 else "sing"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
m = int(sys.argv[4]) if len(sys.argv) > 4 else 150
kw = dict(eps_prox=1e-2, eta_prox=1e-8) if kind == "range" else {}
if kind == "lp":
    qs = [O.generate_lp(n, m, 0, [92, k]) for k in range(N)]
else:
    qs = [O.generate_singular_qp(n, m, 0, (4 * n) // 5, rng=[91, k], in_range=(kind == "range")) for k in range(N)]
b = {k: np.stack([q[k] for q in qs]) for k in ("f", "A", "bupper", "blower", "sense")}
H = None if kind == "lp" else np.stack([q["H"] for q in qs])
for rep in range(3):
    mdl = daqp_amd.BatchModel(N, n, m, 0, **kw)
    t0 = time.time()
    mdl.setup(H, b["f"], b["A"], b["bupper"], b["blower"], b["sense"], init_mask=64)
    t1 = time.time()
    r = mdl.solve()
    t2 = time.time()
    ks, kv = mdl.kernel_ms()
    print(f"rep {rep}: setup {1e3*(t1-t0):.1f} ms (device {ks:.1f}), solve {1e3*(t2-t1):.1f} ms (device {kv:.1f}) -> "
          f"{N/(1e-3*(ks+kv)):.0f} QPs/s device, {N/(t2-t0):.0f} QPs/s with host staging")
    info = mdl.prox_info()
    if rep < 2:
        mdl.close()
print(f"{kind}: N={N} n={n} m={m}: proximal problems {int((info['n_prox'] > 0).sum())}, outer iterations mean {info['outer'].mean():.1f} "
      f"max {info['outer'].max()}, inner iterations mean {r['iter'].mean():.1f} max {r['iter'].max()}, flags "
      f"{dict(zip(*[a.tolist() for a in np.unique(r['exitflag'], return_counts=True)]))}")
ora = O.Oracle()
K = min(N, 128)
st = O.default_settings(**kw)
t0 = time.time()
ref = [ora.quadprog(q.get("H"), q["f"], q["A"], q["bupper"], q["blower"], q["sense"], settings=st) for q in qs[:K]]
t1 = time.time()
same = sum(int(ref[k][4] == r["iter"][k] and ref[k][3] == r["exitflag"][k]) for k in range(K))
dx = max(np.abs(ref[k][0] - r["x"][k]).max() for k in range(K))
print(f"oracle (1 core, strict flags): {K/(t1-t0):.0f} QPs/s; identical iter+flag {same}/{K}; max|dx| {dx:.2e}")
