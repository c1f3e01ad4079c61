"""Cycle probes of the 16-problems-per-wave solve kernel on config C3: python tools/c3_prof.py [N]
(per wave: cycles by phase of the lockstep pass, passes, loop cycles; also the plain rate without probes)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, daqp_amd
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
print(f"C3 N={N}: {N / dt / 1e6:.2f} M QPs/s, kernels setup/solve ms {bm.kernel_ms()}, iters mean {r['iter'].double().mean().item():.2f} max {r['iter'].max().item()}")
bm.enable_profile(True)
r = step(); torch.cuda.synchronize()
print("with probes: solve ms", bm.kernel_ms()[1])
p = bm.read_profile()[:4096]          # one row per persistent wave
p = p[p[:, 11] > 0]
names = ["act", "forward", "b-init+backward", "post(blocking)", "primal", "scan", "decide", "drop / push:row", "push rest", "post-edit", "push:gram"]
passes = p[:, 11].mean()
print(f"waves {len(p)}, passes per wave {passes:.2f} ({p[:, 11].sum() * 16 / N:.2f} per problem), kernel cycles per wave {p[:, 12].mean():.0f} (max {p[:, 12].max()}) = {p[:, 12].mean() / passes:.0f} per pass; "
      f"refill {100 * p[:, 13].sum() / p[:, 12].sum():.1f} %, retire {100 * p[:, 14].sum() / p[:, 12].sum():.1f} %")
for i, nm in enumerate(names):
    print(f"  {nm:18s} {p[:, i].mean() / passes:8.0f} cycles per pass  ({100 * p[:, i].sum() / p[:, 12].sum():5.1f} %)")
