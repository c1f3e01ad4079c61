"""C3 throughput (n=12, m=48, ms=12) quick check: python tools/c3_rate.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daqp_amd
from daqp_amd.synthetic import generate_batch_torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
q = generate_batch_torch(N, 12, 48, 12, 6, seed=43)
bm = daqp_amd.BatchModel(N, 12, 48, 12)
def step():
    bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=64 | 128)
    return bm.solve(out="torch")
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): r = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"C3 N={N}: {N / dt / 1e6:.2f} M QPs/s, kernels setup/solve ms {bm.kernel_ms()}, optimal {(r['exitflag'] == 1).all().item()}, max|x-xref| {(r['x'] - q['xref']).abs().max().item():.2e}")
