"""Default arithmetic mode of the workgroup kernel (inverse factor, fp32-screened scan) on ill-conditioned problems: near-duplicate
active rows at relative distance 1e-9 ... 1e-2 (LDL' pivots down to ~1e-9), n = 65 ... 130, against the oracle.
usage: python tools/inverse_stress.py [trials]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daqp_amd
from oracle import oracle as O
ora = O.Oracle()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 120
os.environ["DAQP_AMD_EXACT"] = "0"
bad_flag = bad_iter = bad_x = 0
worst = 0.0
for trial in range(T):
    rng = np.random.default_rng([299, trial])
    eps = 10.0 ** rng.uniform(-9, -2)
    n = int(rng.integers(65, 131)); m = int(rng.integers(n + 20, 3 * n)); ms = int(rng.integers(0, n // 3))
    na = int(rng.integers(n // 4, n - 4))
    q = O.generate_nasty(n, m, ms, na, eps, rng, n_dup=int(rng.integers(1, 8)), n_eq=0, n_soft=0, dep_eq=False)
    x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    r = ora.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    if flag != r[3]:
        bad_flag += 1; print("flag", trial, n, m, eps, flag, r[3]); continue
    if info["iterations"] != r[4]:
        bad_iter += 1; print("iter", trial, n, m, "eps %.1e" % eps, info["iterations"], r[4], "dx %.1e" % np.abs(x - r[0]).max())
    if flag > 0:
        dx = np.abs(x - r[0]).max() / max(1.0, np.abs(r[0]).max())
        worst = max(worst, dx)
        if dx > 1e-9:
            bad_x += 1; print("x", trial, n, m, "eps %.1e" % eps, "dx %.2e" % dx)
print(f"{T} ill-conditioned problems: exit flag differs {bad_flag}, iteration count differs {bad_iter}, |dx| > 1e-9 (relative to max(1,|x|)) {bad_x}, worst {worst:.2e}")
