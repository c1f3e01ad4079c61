"""What a wave more per CU is worth to the image kernel, apart from hand-overs: C2's shape with FEW active rows at the optimum (working sets that fit
every LDS size tried), the solve launch timed per DAQP_AMD_IMG_ROWS (the LDS carve-up follows it: 42 rows -> 5 workgroups per CU ... 28 -> 8)
and on the full-register kernel.   usage: python tools/img_occupancy.py [N] [nActive]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import daqp_amd
from daqp_amd.synthetic import generate_batch_torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
na = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, m, ms = (int(x) for x in os.environ.get("SHAPE", "50,150,0").split(","))
os.environ["DAQP_AMD_NO_RECHECK"] = "1"
qt = generate_batch_torch(N, n, m, ms, na, seed=42, device="cuda:0")
mask = daqp_amd.UPDATE_unconstrained | daqp_amd.UPDATE_eliminate
ref = None
for label, env in [(f"image kernel, {r} rows", {"DAQP_AMD_IMG_ROWS": str(r)}) for r in [int(x) for x in os.environ.get("SWEEP_ROWS", "42,38,33,28,24").split(",") if x]] + ([] if os.environ.get("SKIP_FULL") else [("full-register kernel", {"DAQP_AMD_NO_IMG32": "1"})]):
    for k in ("DAQP_AMD_IMG_ROWS", "DAQP_AMD_NO_IMG32"):
        os.environ.pop(k, None)
    os.environ.update(env)
    bm = daqp_amd.BatchModel(N, n, m, ms, device=0)
    ts = []
    for it in range(6):
        bm.setup(qt["H"], qt["f"], qt["A"], qt["bupper"], qt["blower"], None, init_mask=mask)
        r = bm.solve(out="torch")
        torch.cuda.synchronize()
        ts.append(bm.kernel_ms()[1])
    it = r["iter"].cpu().numpy()
    if ref is None: ref = it
    print(f"{label}: solve launch {np.median(ts[2:]):.2f} ms per {N}, mean iterations {it.mean():.2f}, same iterations as the first run {np.array_equal(it, ref)}, all optimal {bool((r['exitflag'] == 1).all())}", flush=True)
    bm.close()
