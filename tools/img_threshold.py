"""From which batch size on does the image kernel (two waves per SIMD) beat the full-register kernel (one)?  C2's shape, cold solves and warm steps.
usage: python tools/img_threshold.py [N,N,...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import daqp_amd
from daqp_amd.synthetic import generate_batch_torch
Ns = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [512, 1024, 1536, 2048, 3072, 4096, 8192, 16384]
n, m, ms, na = 50, 150, 0, 20
os.environ["DAQP_AMD_NO_RECHECK"] = "1"
mask = daqp_amd.UPDATE_unconstrained | daqp_amd.UPDATE_eliminate
for N in Ns:
    qt = generate_batch_torch(N, n, m, ms, na, seed=42, device="cuda:0")
    line = f"N = {N:6d}:"
    for label, env in (("image", {"DAQP_AMD_IMG_MIN_BATCH": "1"}), ("full-register", {"DAQP_AMD_NO_IMG32": "1"})):
        for k in ("DAQP_AMD_IMG_MIN_BATCH", "DAQP_AMD_NO_IMG32"):
            os.environ.pop(k, None)
        os.environ.update(env)
        bm = daqp_amd.BatchModel(N, n, m, ms, device=0)
        ts = []
        for it in range(8):
            bm.setup(qt["H"], qt["f"], qt["A"], qt["bupper"], qt["blower"], None, init_mask=mask)
            r = bm.solve(out="torch")
            torch.cuda.synchronize()
            ts.append(bm.kernel_ms()[1])
        bm.setup(qt["H"], qt["f"], qt["A"], qt["bupper"], qt["blower"], None, init_mask=0)
        bm.solve(out="torch")
        gen = torch.Generator(device="cuda:0"); gen.manual_seed(45)
        cur = qt["f"]; wt = []
        for t in range(10):
            cur = cur + 0.05 * torch.randn(cur.shape, generator=gen, dtype=torch.float64, device="cuda:0")
            bm.update(f=cur)
            r = bm.solve(out="torch")
            torch.cuda.synchronize()
            wt.append(bm.kernel_ms()[1])
        line += f"  {label}: cold {np.median(ts[3:]):.3f} ms, warm {np.median(wt):.3f} ms;"
        bm.close()
    print(line, flush=True)
