"""C5 (warm sequence) kernel breakdown via rocprofv3 kernel stats: run under rocprofv3 --kernel-trace --stats"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daqp_amd
from daqp_amd.synthetic import generate_batch_torch
N = 100000
q = generate_batch_torch(N, 50, 150, 0, 20, seed=42)
bm = daqp_amd.BatchModel(N, 50, 150, 0)
bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=0)
bm.solve(out="torch")
g = torch.Generator(device="cuda"); g.manual_seed(45)
f = q["f"].clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(10):
    f += 0.05 * torch.randn(f.shape, generator=g, dtype=torch.float64, device="cuda")
    bm.update(f=f)
    r = bm.solve(out="torch")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"C5: {N * 10 / dt / 1e6:.2f} M warm solves/s, {dt * 100:.2f} ms per step, mean iter {r['iter'].double().mean().item():.2f}, kernel_ms {bm.kernel_ms()}")
