"""Debug aid: compare the HIP path with the oracle stage by stage (LDP arrays, event traces)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
import daqp_amd

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mask = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ora = O.Oracle()
n, m, ms, na, seed, _ = O.CONFIGS[cfg]
q = O.generate_batch(N, n, m, ms, na, seed)
bm = daqp_amd.BatchModel(N, n, m, ms)
bm.enable_trace(4096)
bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], init_mask=mask)
print("setup flags", bm.setup_flags())
oms = []
for k in range(N):
    om = ora.model(n, m, ms)
    om.enable_trace()
    print("oracle setup", om.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None, init_mask=mask))
    oms.append(om)
    for name, a, b in zip(("M", "Rinv", "v", "dupper", "dlower", "scaling"), bm.read_ldp(k), om.ldp()):
        d = np.abs(a - b)
        same = np.array_equal(a.view(np.uint64), b.view(np.uint64))
        print(f"  [{k}] {name:8s} bitwise={same} maxdiff={d.max() if d.size else 0:.3e} nan={np.isnan(a).sum()}")
g = bm.solve()
tr = bm.read_trace()
for k in range(N):
    r = oms[k].solve()
    to = oms[k].get_trace()
    print(f"[{k}] gpu flag {g['exitflag'][k]} iter {g['iter'][k]} | oracle flag {r[3]} iter {r[4]} | dx {np.abs(g['x'][k]-r[0]).max():.3e}")
    L = min(len(to), len(tr[k]))
    diff = np.nonzero(tr[k][:L] != to[:L])[0]
    print("   trace len gpu/oracle", len(tr[k]), len(to), "first diff at", diff[:1], "gpu", tr[k][:12], "ora", to[:12])
print("kernel ms", bm.kernel_ms())
