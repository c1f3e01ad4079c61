"""k_fact_wg (setup_fact.hip.h) against the oracle's LDP and against k_setup's own ordered factorisation: python tools/fact_check.py [n m N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daqp_amd
from oracle import oracle as O
n, m, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (200, 600, 64)
q = O.generate_batch(N, n, m, 0, max(2, n // 3), 44)
ora = O.Oracle()
def ldp_of(env):
    for k, v in env.items(): os.environ[k] = v
    bm = daqp_amd.BatchModel(N, n, m, 0)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
    out = [bm.read_ldp(k) for k in range(min(N, 8))]
    r = bm.solve()
    fl = bm.setup_flags()
    for k in env: os.environ.pop(k)
    return out, r, fl
new, rn, fn = ldp_of({})
old, ro, fo = ldp_of({"DAQP_AMD_NO_FACT_WG": "1"})
worst = 0.0
for k in range(len(new)):
    mod = ora.model(n, m, 0); mod.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k]); ref = mod.ldp()
    for name, a, b_, c in zip(("M", "Rinv", "v", "du", "dl", "scaling"), new[k], old[k], ref):
        sc = max(1e-300, np.abs(c).max())
        e1, e2 = np.abs(a - c).max() / sc, np.abs(b_ - c).max() / sc
        worst = max(worst, e1)
        if k == 0: print(f"  QP 0 {name:8s}: k_fact_wg vs oracle {e1:.2e}   k_setup's own vs oracle {e2:.2e}")
print(f"n={n} m={m}: worst relative deviation of the new LDP from the oracle's over {len(new)} QPs: {worst:.2e}; setup flags equal {(fn == fo).all()}; "
      f"iterations equal {(rn['iter'] == ro['iter']).all()}, exit flags equal {(rn['exitflag'] == ro['exitflag']).all()}, max|dx| {np.abs(rn['x'] - ro['x']).max():.1e}")
