import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, daqp_amd
from oracle import oracle as O
ora = O.Oracle()
for (n, m, ms, na) in ((50, 200, 0, 18), (63, 150, 5, 20), (64, 200, 0, 22), (30, 400, 3, 10), (12, 250, 2, 5)):
    bad = 0
    for k in range(20):
        q = O.generate_qp(n, m, ms, na, rng=[77, n, k])
        x, fval, flag, info = daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        r = ora.quadprog(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
        if not (flag == r[3] and info["iterations"] == r[4] and np.abs(x - r[0]).max() < 1e-9 and np.array_equal(np.sign(info["lam"]), np.sign(r[1]))): bad += 1
    t0 = time.perf_counter()
    for k in range(50): daqp_amd.solve(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    dt = (time.perf_counter() - t0) / 50
    print(f"daqp_quadprog n={n} m={m} ms={ms}: mismatches {bad}/20, {dt*1e3:.3f} ms per call (iterations {info['iterations']})", flush=True)
