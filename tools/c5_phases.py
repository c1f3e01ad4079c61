"""C5 (warm solve) phase probes of the register kernel: prologue (incl. the fused update), loop, epilogue cycles per warm solve."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daqp_amd
from daqp_amd.synthetic import generate_batch_torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
q = generate_batch_torch(N, 50, 150, 0, 20, seed=42)
bm = daqp_amd.BatchModel(N, 50, 150, 0)
bm.enable_profile(True)
bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=0)
bm.solve(out="torch")
g = torch.Generator(device="cuda"); g.manual_seed(45)
f = q["f"].clone()
for t in range(3):
    f += 0.05 * torch.randn(f.shape, generator=g, dtype=torch.float64, device="cuda")
    bm.update(f=f)
    r = bm.solve(out="torch")
torch.cuda.synchronize()
p = bm.read_profile().astype(np.float64)
it = r["iter"].double().mean().item()
print(f"warm solve: mean iterations {it:.2f}, kernel ms {bm.kernel_ms()}")
print("  per warm solve: prologue %d, loop %d, epilogue %d cycles" % tuple(p[:, 28:31].mean(axis=0)[[0, 2, 1]]))
print("  prologue: to end of row loads %d, +to copy issue %d, +to copy done %d" % (p[:, 26].mean(), p[:, 27].mean(), p[:, 31].mean()))
print("  epilogue: issue %d, wait copy %d, x+lam in LDS %d, up to final stores %d" % tuple(p[:, 20:24].mean(axis=0)))
cyc = p[:, :16].sum(axis=0); nit = r["iter"].double().sum().item()
print("  loop/iter: ITER %d (csp %d, blocking %d, primal %d, scan %d), EDIT %d (push %d, drop %d)" % (cyc[1] / nit, cyc[7] / nit, cyc[8] / nit, cyc[9] / nit, cyc[10] / nit, cyc[2] / nit, cyc[13] / nit, cyc[14] / nit))
