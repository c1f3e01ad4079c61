"""What a second wave per SIMD buys THIS state machine: the register kernel's mid shapes (NB*NP <= 32: two waves per SIMD in the shipped
build) timed against a variant of the same code held to one wave per SIMD (DAQP_AMD_LIBRARY = a build with ldp_reg_waves(...) = 1).
usage: python tools/occupancy_probe.py [N] [n,m,nActive;n,m,nActive;...]      (run once per library; prints one line per shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daqp_amd
from daqp_amd.synthetic import generate_batch_torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
SHAPES = [tuple(int(v) for v in t.split(",")) for t in sys.argv[2].split(";")] if len(sys.argv) > 2 else [(30, 120, 12), (32, 64, 12), (24, 128, 10)]
for (n, m, na) in SHAPES:
    q = generate_batch_torch(N, n, m, 0, na, 7000 + n)
    bm = daqp_amd.BatchModel(N, n, m, 0)
    best = None
    for rep in range(4):
        bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
        r = bm.solve(out="torch")
        torch.cuda.synchronize()
        ks, kl = bm.kernel_ms()
        best = kl if best is None else min(best, kl)
    it = r["iter"].double().mean().item()
    ok = bool((r["exitflag"] == 1).all().item())
    print(f"lib={os.path.basename(os.environ.get('DAQP_AMD_LIBRARY', 'default'))} shape n={n} m={m} nActive={na} N={N}: solve launch {best:.3f} ms, "
          f"mean iterations {it:.2f}, all optimal {ok}, {best * 1e-3 * 1024 * 2.4e9 / (N * it):.0f} SIMD-cycles per iteration (1024 SIMDs x 2.4 GHz x time / iterations)", flush=True)
    bm.close()
os._exit(0)
