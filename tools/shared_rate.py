"""Shared-structure (condensed MPC) throughput: ONE H, A (C2 shape), N problems with their own f / bounds.
Reports the first (cold) solves/s after setup_shared and warm solves/s over T update(f)+solve steps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daqp_amd
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n, m, ms, na = 50, 150, 0, 20
q0 = O.generate_qp(n, m, ms, na, rng=[811, n])
g = torch.Generator(device="cuda"); g.manual_seed(5)
dd = dict(dtype=torch.float64, device="cuda")
H = torch.from_numpy(q0["H"]).cuda(); A = torch.from_numpy(q0["A"]).cuda()
f = torch.from_numpy(q0["f"]).cuda()[None, :] + 0.3 * torch.randn((N, n), generator=g, **dd)
sh = 0.05 * torch.randn((N, m), generator=g, **dd)
bu = torch.from_numpy(q0["bupper"]).cuda()[None, :] + sh; bl = torch.from_numpy(q0["blower"]).cuda()[None, :] + sh
bm = daqp_amd.BatchModel(N, n, m, ms)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bm.setup_shared(H, f, A, bu, bl)
    r = bm.solve(out="torch")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"shared C2 shape, N={N}: setup_shared + cold solve {N / dt / 1e6:.2f} M QPs/s ({dt * 1e3:.2f} ms, mean iter {r['iter'].double().mean().item():.1f}, "
      f"optimal {(r['exitflag'] == 1).double().mean().item():.4f})")
T = 10
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T):
    f += 0.05 * torch.randn(f.shape, generator=g, **dd)
    bm.update(f=f)
    r = bm.solve(out="torch")
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"  warm: {N * T / dt / 1e6:.2f} M solves/s ({dt / T * 1e3:.2f} ms per step, mean iter {r['iter'].double().mean().item():.2f}), kernel_ms {bm.kernel_ms()}")
if len(sys.argv) > 2:
    bm.enable_profile(True)
    f += 0.05 * torch.randn(f.shape, generator=g, **dd); bm.update(f=f); r = bm.solve(out="torch"); torch.cuda.synchronize()
    p = bm.read_profile().astype(float)
    print("  per QP cycles: prologue %d (rows %d, +copy issue %d, +copy done %d), epilogue %d, loop %d; iterations %.2f"
          % (p[:, 28].mean(), p[:, 26].mean(), p[:, 27].mean(), p[:, 31].mean(), p[:, 29].mean(), p[:, 30].mean(), r["iter"].double().mean().item()))
