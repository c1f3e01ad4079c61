"""C4 throughput (n=200, m=600) on a reduced batch + parity sample: python tools/c4_rate.py [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daqp_amd
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n, m, ms, na, seed, _ = O.CONFIGS["C4"]
na = int(os.environ.get("C4_NA", na))      # (experiments: fewer rows active at the optimum -> smaller working sets)
if os.environ.get("C4_SHAPE"): n, m, na = (int(v) for v in os.environ["C4_SHAPE"].split(","))   # (experiments: other workgroup-kernel shapes, "n,m,n_active")
from daqp_amd.synthetic import generate_batch_torch
q = generate_batch_torch(N, n, m, ms, na, seed=seed)
qn = {k: q[k][:16].cpu().numpy() for k in ("H", "f", "A", "bupper", "blower")}
bm = daqp_amd.BatchModel(N, n, m, ms)
if len(sys.argv) > 2: bm.enable_profile(True)
def step():
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=64 | 128)
    return bm.solve(out="torch")
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); r = step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
ref = O.Oracle().quadprog_batch(qn["H"][:16], qn["f"][:16], qn["A"][:16], qn["bupper"][:16], qn["blower"][:16], None, ms=ms)
ok = (np.sign(r["lam"][:16].cpu().numpy()) == np.sign(ref[1])).all() and (r["iter"][:16].cpu().numpy() == ref[4]).all()
if len(sys.argv) > 2:
    pr = bm.read_profile().astype(float)
    it = r["iter"].double().sum().item()
    names = ["csp", "ratio test(no block)", "primal", "scan", "add (incl. guard)", "ratio test + drop", "-", "-"]
    sp = pr[:, 16:22].mean(axis=0)
    print("  setup cycles per QP: checks %.0f, Cholesky %.0f, inverse %.0f, v/x_unc + unnormalised M %.0f, normalise + d %.0f, bounds+write-back %.0f | total %.0f" % (*sp, sp.sum()))
    if os.environ.get("DAQP_AMD_NO_FACT_WG"): print("  matrix-core phase: A operand loads %.0f, R^-1 loads + mfma %.0f, blocked stores %.0f" % tuple(pr[:, 22:25].mean(axis=0)))
    else: print("  k_fact_wg cycles per QP (a workgroup each): load + symmetrise %.0f, Cholesky %.0f, inverse + stores %.0f" % tuple(pr[:, 22:25].mean(axis=0)))
    print("  screening scans per QP %.1f, fp64 re-scans %.2f; per screening scan: master's own part %.0f cycles, its wait for the other waves %.0f" % (pr[:, 25].mean(), pr[:, 26].mean(), pr[:, 27].sum() / pr[:, 25].sum(), pr[:, 28].sum() / pr[:, 25].sum()))
    print("  cycles/iteration:", ", ".join(f"{names[k]} {pr[:, k].sum() / it:.0f}" for k in range(6)), f"| total {pr[:, :6].sum() / it:.0f}")
    if not os.environ.get("DAQP_AMD_NO_WG"):   # workgroup kernel: finer split (the setup kernel's counters sit at [16:22])
        n2 = ["csp forward", "csp backward", "add: fetch+gram", "add: chains", "drop: compaction", "drop: chain", "scan: |u|^2 + pick"]
        print("  of which:", ", ".join(f"{n2[k]} {pr[:, 6 + k].sum() / it:.0f}" for k in range(7)), f"| general forward sweeps per iteration {pr[:, 15].sum() / it:.3f} | backward: loads {pr[:, 13].sum() / it:.0f}, chains {pr[:, 14].sum() / it:.0f}")
if len(sys.argv) > 2 and not os.environ.get("DAQP_AMD_NO_WG"):
    adds = pr[:, 25].sum()
    print("  per append (cycles): row fetch until it is in LDS %.0f, Gram column after that %.0f, W g / l / new row of W %.0f" % (pr[:, 31].sum() / adds, (pr[:, 29].sum() - pr[:, 31].sum()) / adds, pr[:, 30].sum() / adds))
print(f"n={n} m={m} active {na}, N={N}: {N / dt:.0f} QPs/s, kernels setup/solve ms {bm.kernel_ms()}, mean iter {r['iter'].double().mean().item():.1f}, optimal {(r['exitflag'] == 1).all().item()}, parity(16) {ok}, max|dx| {np.abs(r['x'][:16].cpu().numpy() - ref[0]).max():.1e}")
