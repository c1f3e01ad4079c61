import os, sys
sys.path.insert(0, "/root/repo")
import torch, daqp_amd
from daqp_amd.synthetic import generate_batch_torch
N = 20000
for (n, m) in ((50, 193), (50, 256), (40, 256), (26, 256), (63, 150), (63, 192), (56, 129), (30, 300), (32, 512), (50, 300), (50, 384), (63, 300), (60, 320)):
    q = generate_batch_torch(N, n, m, 0, max(2, n // 3), 8000 + n)
    out = []
    for off in ("", "1"):
        if off: os.environ["DAQP_AMD_NO_IMG_ONLY"] = "1"
        else: os.environ.pop("DAQP_AMD_NO_IMG_ONLY", None)
        bm = daqp_amd.BatchModel(N, n, m, 0)
        best = None
        for rep in range(3):
            bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
            r = bm.solve(out="torch"); torch.cuda.synchronize()
            kl = bm.kernel_ms()[1]; best = kl if best is None else min(best, kl)
        out.append((best, r["iter"].double().mean().item(), bool((r["exitflag"] == 1).all().item())))
        bm.close()
    print(f"n={n} m={m}: image alone {out[0][0]:.2f} ms vs M streamed {out[1][0]:.2f} ms per {N} (mean iter {out[0][1]:.1f}/{out[1][1]:.1f}, optimal {out[0][2]}/{out[1][2]})", flush=True)
