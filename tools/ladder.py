"""The reference's own benchmark ladder (interfaces/daqp-julia/test/benchmark.jl:36-40: (n, m, ms, nActive), kappa = 100) as batches on the
GPU: setup + solve per QP with inputs resident in HBM, both arithmetic modes, the reference library on 32 host threads next to it
(daqp_quadprog, oracle/_ref, a bounded sample) and the parity of that sample.     usage: python tools/ladder.py [scale]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daqp_amd
from oracle import oracle as O

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
LADDER = [((10, 50, 5, 8), 200000, 8192), ((50, 250, 25, 40), 40000, 2048), ((100, 500, 50, 80), 8000, 512),
          ((200, 1000, 100, 160), 2048, 64), ((500, 2500, 250, 400), 256, 8)]
ref = os.path.join(O.HERE, "_ref", "libdaqp_ref.so")
for (n, m, ms, na), N, S in LADDER:
    N = max(S, int(N * scale))
    qn = O.generate_batch(min(N, 4 * S), n, m, ms, na, 9000 + n)          # (numpy generator: the parity tests' one), tiled to the batch size
    reps = (N + qn["f"].shape[0] - 1) // qn["f"].shape[0]
    q = {k: torch.from_numpy(np.ascontiguousarray(np.concatenate([qn[k]] * reps)[:N])).cuda() for k in ("H", "f", "A", "bupper", "blower")}
    line = f"(n, m, ms, nActive) = ({n}, {m}, {ms}, {na}), N = {N}:"
    for exact in (0, 1):
        os.environ["DAQP_AMD_EXACT"] = str(exact)
        bm = daqp_amd.BatchModel(N, n, m, ms)
        def step():
            bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=64 | 128)
            return bm.solve(out="torch")
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        line += f" {'exact' if exact else 'default'} {N / dt:,.0f} QPs/s (kernels {bm.kernel_ms()[0]:.2f} + {bm.kernel_ms()[1]:.2f} ms)"
        if not exact:
            rd = {k: r[k][:S].cpu().numpy() for k in ("x", "lam", "iter", "exitflag")}
            it = r["iter"].double().mean().item()
        bm.close()
    os.environ.pop("DAQP_AMD_EXACT")
    cpu_dt, x, lam, fval, fl, itc = O.timed_cpu_batch(ref, 32, *(qn[k][:S] for k in ("H", "f", "A", "bupper", "blower")), ms=ms, passes=1)
    same = bool((rd["iter"] == itc).all() and (rd["exitflag"] == fl).all() and (np.sign(rd["lam"]) == np.sign(lam)).all())
    line += f" | mean iterations {it:.1f} | reference on 32 threads {S / cpu_dt:,.0f} QPs/s; sample of {S}: flags, iterations, active sets identical {same}, max|dx| {np.abs(rd['x'] - x).max():.1e}"
    print(line, flush=True)
