"""k_setup_blk with simple bounds (ms > 0) against k_setup_fast (DAQP_AMD_NO_BLK_BOUNDS=1): setup launch per 100 000 QPs, and the LDP both leave
(rows < ms of M, scaling, d, packed R^-1) compared entry by entry.   usage: python tools/blk_bounds_rate.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, daqp_amd
from daqp_amd.synthetic import generate_batch_torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for (n, m, ms, na) in ((50, 150, 50, 20), (50, 150, 10, 20), (40, 100, 40, 14), (63, 128, 20, 22), (24, 80, 24, 8)):
    q = generate_batch_torch(N, n, m, ms, na, 4242)
    res = {}
    for off in ("", "1"):
        if off: os.environ["DAQP_AMD_NO_BLK_BOUNDS"] = "1"
        else: os.environ.pop("DAQP_AMD_NO_BLK_BOUNDS", None)
        bm = daqp_amd.BatchModel(N, n, m, ms)
        best = None
        for rep in range(4):
            bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=64 | 128)
            r = bm.solve(out="torch"); torch.cuda.synchronize()
            ks, kl = bm.kernel_ms(); best = (ks, kl) if best is None or ks < best[0] else best
        ldp = [bm.read_ldp(k) for k in (0, 1, N - 1)]
        res[off] = (best, r["iter"].cpu().numpy(), r["exitflag"].cpu().numpy(), r["x"].cpu().numpy(), ldp)
        bm.close()
    a, b = res[""], res["1"]
    same = np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    dmax = max(np.abs(x - y).max() / max(1.0, np.abs(y).max()) for la, lb in zip(a[4], b[4]) for x, y in zip(la, lb))
    print(f"n={n} m={m} ms={ms}: setup k_setup_blk {a[0][0]:.2f} ms vs k_setup_fast {b[0][0]:.2f} ms per {N} (solve {a[0][1]:.2f} / {b[0][1]:.2f}); iterations and flags identical {same}, "
          f"max |dx| {np.abs(a[3] - b[3]).max():.1e}, LDP max rel diff {dmax:.1e}", flush=True)
