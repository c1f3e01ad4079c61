import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, daqp_amd
from oracle import oracle as O
from daqp_amd.synthetic import generate_batch_torch
N = 2048
n, m, ms, na, seed, _ = O.CONFIGS["C4"]
q = generate_batch_torch(N, n, m, ms, na, seed=seed)
bm = daqp_amd.BatchModel(N, n, m, ms)
bm.enable_profile(True)
bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=192)
r = bm.solve(out="torch")
torch.cuda.synchronize()
pr = bm.read_profile().astype(float)
it = r["iter"].double().sum().item(); adds = pr[:, 25].sum()
print("PROBE2 per append: W g rows %.0f, -l'W cols %.0f | per iteration (CSP): rows %.0f, W'z cols %.0f" % (pr[:, 29].sum() / adds, pr[:, 30].sum() / adds, pr[:, 31].sum() / it, pr[:, 28].sum() / it))
