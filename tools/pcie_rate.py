"""PCIe-inclusive rate: daqp_quadprog_batch on HOST buffers (the library stages H2D, solves, copies back)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import daqp_amd
from daqp_amd.synthetic import generate_batch_torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
q = generate_batch_torch(N, 50, 150, 0, 20, seed=42)
h = {k: q[k].cpu().numpy() for k in ("H", "f", "A", "bupper", "blower")}
for rep in range(3):
    t0 = time.perf_counter()
    r = daqp_amd.solve_batch(h["H"], h["f"], h["A"], h["bupper"], h["blower"], None, ms=0)
    dt = time.perf_counter() - t0
    print(f"host buffers, N={N}: {N / dt:,.0f} QPs/s ({dt * 1e3:.1f} ms, {(h['H'].nbytes + h['A'].nbytes) / dt / 1e9:.1f} GB/s of input), all optimal {(r['exitflag'] == 1).all()}")
