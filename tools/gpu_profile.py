"""Phase breakdown of the solve kernel (cycle counters) + kernel times on a mid-size batch."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import daqp_amd
from daqp_amd.synthetic import generate_batch_torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n, m, ms, na = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (50, 150, 0, 20)
q = generate_batch_torch(N, n, m, ms, na, seed=42)
for prof in (False, True):
    bm = daqp_amd.BatchModel(N, n, m, ms)
    if prof:
        bm.enable_profile(True)
    for rep in range(3):
        bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=64)
        torch.cuda.synchronize()
        if prof and rep == 2:
            ps = bm.read_profile().astype(np.float64).mean(axis=0)
            print("  setup cycles per QP: chol %d, inverse+v %d, xunc %d, M %d, norm/d/store %d, tail %d, total %d"
                  % (ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], ps[:10].sum()))
            print("    M split: bounds/pack/frags %d, staging %d, mfma+writeback %d, (slot 3 = row load); slot 9 (k_setup_blk: checks + H into tiles, before the factorisation) %d" % (ps[6], ps[7], ps[8], ps[9]))
        res = bm.solve(out="torch")
        torch.cuda.synchronize()
    su, so = bm.kernel_ms()
    it = res["iter"].cpu().numpy()
    print(f"profiling={prof}: N={N} k_setup {su:.2f} ms, k_ldp {so:.2f} ms, mean iter {it.mean():.1f}, "
          f"all optimal {(res['exitflag'] == 1).all().item()}, max|x-xref| {(res['x'] - q['xref']).abs().max().item():.2e}")
    if prof:
        p = bm.read_profile().astype(np.float64)
        # profile slots per QP: [0:7] cycles per state, [7:11] ITER parts, [11:13] csp fwd/bwd, [13:15] add/remove,
        # [16:23] visits per state, [20:24] epilogue parts (overlaying the never-visited ACT states' visit slots),
        # [24:26] remove parts, [26],[27],[31] prologue parts, [28:31] prologue/epilogue/loop totals
        names = ["START", "ITER(csp..scan..commit)", "EDIT(add/drop+pivot+guard)", "ACT_BEGIN", "ACT_NEXT", "ACT_POST", "DONE"]
        cyc, vis = p[:, :16].sum(axis=0), p[:, 16:].sum(axis=0)
        nit = it.sum()
        print("  state: cycles/iteration (cycles/visit, visits/iteration)")
        for k in range(3):
            if vis[k] > 0:
                print(f"    {names[k]:24s} {cyc[k] / nit:9.0f}  ({cyc[k] / vis[k]:8.0f}, {vis[k] / nit:5.2f})")
        print("    total/iter", round(cyc[:7].sum() / nit), " | csp fwd %d bwd %d (cycles/iter)" % tuple(cyc[11:13] / nit))
        print("    ITER parts/iter: csp %d, blocking %d, primal %d, scan %d | EDIT parts/iter: push %d, drop %d"
              % (cyc[7] / nit, cyc[8] / nit, cyc[9] / nit, cyc[10] / nit, cyc[13] / nit, cyc[14] / nit))
        print("    scans left undecided by the fp32 image (image kernel only): %d of %d iterations" % (cyc[15], nit))
        print("    per QP: prologue %d, epilogue %d, loop %d cycles" % tuple(p[:, 28:31].mean(axis=0)))
        print("    prologue: to end of row loads %d, +to copy issue %d, +to copy done %d" % (p[:, 26].mean(), p[:, 27].mean(), p[:, 31].mean()))
        print("    epilogue: issue %d, wait copy %d, x+lam in LDS %d, up to final stores %d" % tuple(p[:, 20:24].mean(axis=0)))
        pc = p[:, 24:28].sum(axis=0) / nit
        print("    drop parts/iter: compaction %d, C1 update %d (of drop total above)" % tuple(pc[:2]))
    bm.close()
