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
# the multi-device entry on the same host buffers (one device: the shards' pinned, chunked, double-buffered staging against the
# single-device call's pageable copies; the box has one GPU, so "two shards" means two streams sharing it)
for devs in ([0], [0, 0]):
    for rep in range(2):
        t0 = time.perf_counter()
        r = daqp_amd.solve_batch_multi(h["H"], h["f"], h["A"], h["bupper"], h["blower"], None, ms=0, devices=devs)
        dt = time.perf_counter() - t0
    print(f"daqp_quadprog_batch_multi, devices {devs}, N={N}: {N / dt:,.0f} QPs/s ({dt * 1e3:.1f} ms, {(h['H'].nbytes + h['A'].nbytes) / dt / 1e9:.1f} GB/s of input), all optimal {(r['exitflag'] == 1).all()}")
mb = daqp_amd.MultiBatchModel(N, 50, 150, 0, devices=[0])
for rep in range(2):
    t0 = time.perf_counter()
    mb.setup(h["H"], h["f"], h["A"], h["bupper"], h["blower"], None, init_mask=192)
    t1 = time.perf_counter()
    r = mb.solve()
    t2 = time.perf_counter()
print(f"persistent multi-device batch (workspaces kept), devices [0]: setup incl. staging {1e3 * (t1 - t0):.1f} ms, solve incl. results {1e3 * (t2 - t1):.1f} ms = {N / (t2 - t0):,.0f} QPs/s")
mb.close()
