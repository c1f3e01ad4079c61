"""One workspace (single-problem symbols) through every kind of daqp_update_ldp mask in the DEFAULT arithmetic mode, step by step against the
oracle: python tools/ho_debug.py n,m,ms,nActive   (env: DAQP_AMD_REG_ROWS=<k> forces the register kernel hand-over, DAQP_AMD_NO_REG_HANDOVER=1
turns it off).  The iterate of an INFEASIBLE exit is no solution: there only flag / iterations / working set are comparable."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
os.environ["DAQP_AMD_EXACT"] = "0"; os.environ["DAQP_AMD_NO_RECHECK"] = "1"
import daqp_amd
from oracle import oracle as O
import test_gpu_hand_over as T
ora = O.Oracle()
n, m, ms, na = (int(v) for v in sys.argv[1].split(","))
q = O.generate_qp(n, m, ms, na, rng=[3700 + n, 0])
mdl = daqp_amd.Model(); mdl.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
om = ora.model(n, m, ms); om.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
cur = dict(q)
for step, mask in enumerate([0] + T.MASKS):
    if mask:
        kw = T.perturbed(cur, mask, np.random.default_rng([53, 0, step]), n, m, ms); cur.update(kw)
        a, b = mdl.update_mask(mask, **kw), om.update(mask, **kw)
    x, fval, gflag, info = mdl.solve(); r = om.solve()
    print(f"step {step} mask {mask:2d}: flag {gflag}/{r[3]} iter {info['iterations']}/{r[4]} same active set {np.array_equal(np.sign(info['lam']), np.sign(r[1]))} dx {np.abs(x - r[0]).max():.2e} dlam {np.abs(info['lam'] - r[1]).max():.2e}")
