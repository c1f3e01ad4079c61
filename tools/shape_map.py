"""Where the steps in the shape space are: solve launch per QP and SIMD-cycles per iteration over a grid of (n, m), with the kernel family that
serves each shape (the dispatch rule of daqp_batch_create restated: register shapes (NB, NP) for working sets of <= 64 rows -- 65 on (2,32) --,
the fp32-image kernels of the (3,25) and (2,32) shapes for batches of >= 10 240 problems, the workgroup kernel for 65 .. 256 rows, otherwise the one-wave generic kernel with M streamed).  nActive = n / 3.
usage: python tools/shape_map.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daqp_amd
from daqp_amd.synthetic import generate_batch_torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
REG = [(1, 6), (1, 8), (1, 13), (1, 16), (2, 16), (3, 8), (4, 8), (1, 25), (3, 25), (2, 32)]      # kRegShapes of daqp_amd.hip, first fit
WAVES = {(1, 6): 4, (1, 8): 4, (1, 13): 3, (1, 16): 2, (2, 16): 2, (3, 8): 2, (4, 8): 2, (1, 25): 2, (3, 25): 1, (2, 32): 1}


def family(n, m):
    cap, nblk, npair = n + 1, (m + 63) // 64, (n + 1) // 2
    if cap <= 65:
        for nb, np_ in REG:
            if nblk <= nb and npair <= np_:
                if cap <= 64 or (nb, np_) == (2, 32):
                    if cap <= 64 and (nb, np_) in ((3, 25), (2, 32)) and N >= 10240:      # (round 6: the fp32-image kernels, two waves per SIMD, batches of >= 10 240)
                        return f"image({nb},{np_}) x2"
                    if cap > 64:
                        return "image(4,32) x1, image alone"      # (n = 64: the default arithmetic's batch solves; (2,32) registers for the rest)
                    return f"reg({nb},{np_}) x{WAVES[(nb, np_)]}"
                break
    if n > 16 and n <= 64:
        for nb, np_ in ((4, 32), (8, 16), (6, 25), (5, 32)):
            if nblk <= nb and npair <= np_:
                return f"image({nb},{np_}) x1, image alone"
    if 64 < cap <= 256:
        return "workgroup"
    return "generic (M streamed)"


print(f"N = {N} QPs per shape, default arithmetic, nActive = n/3; columns: n m | kernel family | solve launch ms | us per QP | mean iterations | SIMD-cycles per iteration")
for n in (8, 12, 16, 17, 26, 27, 32, 33, 40, 50, 51, 56, 63, 64, 65):
    for m in (32, 64, 65, 128, 129, 150, 192, 193, 256):
        if m <= n:
            continue
        try:
            q = generate_batch_torch(N, n, m, 0, max(2, n // 3), 8000 + n)
            bm = daqp_amd.BatchModel(N, n, m, 0)
            best = None
            for rep in range(3):
                bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
                r = bm.solve(out="torch")
                torch.cuda.synchronize()
                ks, kl = bm.kernel_ms()
                best = kl if best is None else min(best, kl)
            it = r["iter"].double().mean().item()
            ok = bool((r["exitflag"] == 1).all().item())
            print(f"{n:3d} {m:3d} | {family(n, m):28s} | {best:8.3f} | {best * 1e3 / N:7.3f} | {it:6.2f} | {best * 1e-3 * 1024 * 2.4e9 / (N * it):8.0f}{'' if ok else '  (not all optimal)'}", flush=True)
            bm.close()
            del q, r
        except Exception as e:   # (the torch generator of this tool, not the library: hipBLAS workspace at large n x N)
            print(f"{n:3d} {m:3d} | {family(n, m):28s} | skipped: {str(e)[:80]}", flush=True)


# ---- the workgroup kernel's shapes (round 6): four-wave workgroups two per CU where the whole factor fits half the LDS, the tiered launch
# (two per CU, the factor's tail in HBM) for cold solves of the four-chunk shapes beyond that
NL = int(os.environ.get("SHAPE_MAP_LARGE_N", "4096"))
print(f"\nworkgroup-kernel shapes, N = {NL} QPs per shape, nActive = n/3; columns: n m | residency | setup ms | solve launch ms | us per QP | mean iterations")
for n in (72, 80, 100, 110, 120, 128, 150, 200, 229):
    for m in (200, 300, 400, 600):
        if m <= n + 8:
            continue
        try:
            cap = n + 1
            from_lds = 8 * ((8 * (128 if cap <= 128 else 256) + 516 + 136 * 8 + 24) + ((5 * (128 if cap <= 128 else 256) + 16 + (m + 3) // 4 * 4 + 3) // 4 * 4) // 2 + (cap * (cap + 1) // 2 + 1) // 2 * 2)
            res = "4 waves x 2 per CU" if from_lds <= (160 * 1024 - 512) // 2 - 256 else ("tiered: 4 waves x 2 per CU" if NL >= 512 else "up to 8 waves x 1 per CU")
            q = generate_batch_torch(NL, n, m, 0, max(2, n // 3), 8100 + n)
            bm = daqp_amd.BatchModel(NL, n, m, 0)
            best = None
            for rep in range(3):
                bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None)
                r = bm.solve(out="torch")
                torch.cuda.synchronize()
                ks, kl = bm.kernel_ms()
                best = (ks, kl) if best is None or kl < best[1] else best
            it = r["iter"].double().mean().item()
            ok = bool((r["exitflag"] == 1).all().item())
            print(f"{n:3d} {m:3d} | {res:28s} | {best[0]:7.2f} | {best[1]:8.2f} | {best[1] * 1e3 / NL:7.2f} | {it:6.1f}{'' if ok else '  (not all optimal)'}", flush=True)
            bm.close()
            del q, r
        except Exception as e:
            print(f"{n:3d} {m:3d} | skipped: {str(e)[:80]}", flush=True)
