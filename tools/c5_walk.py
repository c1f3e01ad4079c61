"""C5's warm walk over MANY steps (bench.py walks f for (warmup + steps) x 10 steps): solve launch ms and mean iterations along the walk.
usage: python tools/c5_walk.py [N] [steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daqp_amd
from daqp_amd.synthetic import generate_batch_torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 120
q = generate_batch_torch(N, 50, 150, 0, 20, seed=42)
bm = daqp_amd.BatchModel(N, 50, 150, 0)
bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=0)
bm.solve(out="torch")
g = torch.Generator(device="cuda"); g.manual_seed(45)
f = q["f"].clone()
out = []
for t in range(S):
    f += 0.05 * torch.randn(f.shape, generator=g, dtype=torch.float64, device="cuda")
    bm.update(f=f)
    r = bm.solve(out="torch")
    torch.cuda.synchronize()
    if t in (0, 4, 9, 19, 39, 59, 79, 99, 119, S - 1):
        na = (r["lam"] != 0).sum(dim=1).double()
        out.append(f"step {t + 1}: solve launch {bm.kernel_ms()[1]:.2f} ms, mean iterations {r['iter'].double().mean().item():.2f}, active rows mean {na.mean().item():.1f} max {int(na.max().item())}, all optimal {bool((r['exitflag'] == 1).all())}")
print("\n".join(out))
