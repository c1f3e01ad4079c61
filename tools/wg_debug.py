import os, sys
sys.path.insert(0, '/root/repo')
os.environ["DAQP_AMD_NO_RECHECK"] = "1"
import numpy as np, daqp_amd
from oracle import oracle as O
for (n, m, ms, na) in [(70, 160, 5, 25), (200, 600, 0, 80), (130, 300, 0, 50)]:
    q = O.generate_batch(4, n, m, ms, na, 4242 + n)
    ref = O.Oracle().quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    bm = daqp_amd.BatchModel(4, n, m, ms)
    bm.enable_trace(4096)
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=192)
    g = bm.solve()
    tr = bm.read_trace()
    print((n, m), "gpu flag", g["exitflag"], "iter", g["iter"], "ref", ref[3], ref[4], "trace0", tr[0][:12])
