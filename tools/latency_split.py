import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import daqp_amd
from oracle import oracle as O
for cfg in ("C2", "C1", "C2", "C1"):
    n, m, ms, na, seed, _ = O.CONFIGS[cfg]
    q = O.generate_qp(n, m, ms, na, rng=[seed, 0])
    args = (q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    mdl = daqp_amd.Model(); mdl.setup(*args)
    for _ in range(5): mdl.update(f=q["f"]); mdl.solve()
    K = 200
    tu = ts = 0
    for _ in range(K):
        t0 = time.perf_counter(); mdl.update(f=q["f"]); t1 = time.perf_counter(); r = mdl.solve(); t2 = time.perf_counter()
        tu += t1 - t0; ts += t2 - t1
    print(cfg, "sense", np.unique(q["sense"]), f"update {1e3*tu/K:.3f} ms, solve {1e3*ts/K:.3f} ms", r[2])
