"""Parity at BASELINE.json's FULL sizes on SURVEY 8(d)'s own draws (QP k of a config = default_rng([seed, k]), k = 0 .. N-1; the warm
walk of C5 = default_rng([45, k, t])): the default-mode HIP path against the reference library itself (oracle/_ref/libdaqp_ref.so, the
reference's release flags) run on the box's host threads by oracle/ref_batch.c.  Per config: fraction of QPs with identical exit flag,
iteration count and active set (index and side), max |dx|, and what both sides took.
usage: python tools/full_size_parity.py [C2,C3,C4,C5] [scale] [exact]     (scale < 1 shrinks every N: dry runs; exact: the exact
arithmetic mode against the reference's STRICT build (-O2 -ffp-contract=off), x and lam compared bit for bit as well)"""
import os, sys, time, json
import multiprocessing as mp
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):      # the generator pool forks one process per core: a threaded BLAS inside each
    os.environ.setdefault(_v, "1")                                              # of them (n = 200: QR of every problem) oversubscribes the box a hundredfold
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

CHUNK = {"C2": 25_000, "C3": 125_000, "C4": 2_500, "C5": 25_000}     # host memory per chunk <= ~4 GB
T_WARM = 10


def _gen(args):
    cfg, lo, cnt = args
    n, m, ms, na, seed, _ = O.CONFIGS["C2" if cfg == "C5" else cfg]
    return O.generate_batch(cnt, n, m, ms, na, seed, start=lo)


def generate(pool, cfg, lo, cnt, parts):
    edges = np.linspace(0, cnt, parts + 1).astype(int)
    outs = pool.map(_gen, [(cfg, lo + int(edges[i]), int(edges[i + 1] - edges[i])) for i in range(parts) if edges[i + 1] > edges[i]])
    return {k: np.concatenate([o[k] for o in outs]) for k in outs[0]}


def _walk(args):
    lo, cnt, n, f0 = args
    fs = np.empty((T_WARM, cnt, n))
    f = f0.copy()
    for t in range(T_WARM):
        for k in range(cnt):
            f[k] = f[k] + 0.05 * np.random.default_rng([45, lo + k, t]).standard_normal(n)
        fs[t] = f
    return fs


def walk(pool, lo, f0, parts):
    cnt, n = f0.shape
    edges = np.linspace(0, cnt, parts + 1).astype(int)
    outs = pool.map(_walk, [(lo + int(edges[i]), int(edges[i + 1] - edges[i]), n, f0[edges[i]:edges[i + 1]]) for i in range(parts)
                            if edges[i + 1] > edges[i]])
    return np.concatenate(outs, axis=1)


def main():
    cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["C2", "C3", "C4", "C5"]
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    exact = len(sys.argv) > 3 and sys.argv[3] == "exact"
    threads = os.cpu_count() or 8
    pool = mp.get_context("fork").Pool(min(threads, 32))          # forked BEFORE the HIP runtime is touched
    import daqp_amd
    os.environ["DAQP_AMD_NO_RECHECK"] = "1"                         # the default-mode kernels' own verdicts
    os.environ["DAQP_AMD_EXACT"] = "1" if exact else "0"
    if not O.reference_available(strict=exact):
        raise SystemExit("oracle/_ref/libdaqp_ref*.so did not travel: build it here (python __graft_entry__.py) first")
    refname = "libdaqp_ref_strict.so" if exact else "libdaqp_ref.so"
    ref = os.path.join(O.HERE, "_ref", refname)
    report = {"threads": threads, "mode": "exact" if exact else "default",
              "reference": f"oracle/_ref/{refname} ({'-O2 -ffp-contract=off' if exact else 'release flags'}), oracle/ref_batch.c driver", "configs": {}}
    for cfg in cfgs:
        n, m, ms, na, seed, Nfull = O.CONFIGS["C2" if cfg == "C5" else cfg]
        N = max(1, int(Nfull * scale))
        acc = dict(N=0, flag=0, it=0, aset=0, bits=0, dx=0.0, gpu_s=0.0, cpu_s=0.0, gen_s=0.0, iters=0, not_optimal=0, dxref=0.0)
        for lo in range(0, N, CHUNK[cfg]):
            cnt = min(CHUNK[cfg], N - lo)
            t0 = time.perf_counter()
            q = generate(pool, cfg, lo, cnt, 4 * pool._processes)
            fs = walk(pool, lo, q["f"], 4 * pool._processes) if cfg == "C5" else None
            acc["gen_s"] += time.perf_counter() - t0
            if cfg != "C5":
                t0 = time.perf_counter()
                g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
                acc["gpu_s"] += time.perf_counter() - t0
                dt, x, lam, fval, flag, it = O.timed_cpu_batch(ref, threads, q["H"], q["f"], q["A"], q["bupper"], q["blower"], ms)
                acc["cpu_s"] += dt
                gx, glam, gflag, git = g["x"][None], g["lam"][None], g["exitflag"][None], g["iter"][None]
                x, lam, flag, it = x[None], lam[None], flag[None], it[None]
                acc["dxref"] = max(acc["dxref"], float(np.abs(g["x"] - q["xref"]).max()))
            else:
                t0 = time.perf_counter()
                bm = daqp_amd.BatchModel(cnt, n, m, ms)
                bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"])
                bm.solve()
                gx, glam = np.empty((T_WARM, cnt, n)), np.empty((T_WARM, cnt, m))
                gflag, git = np.empty((T_WARM, cnt), np.int32), np.empty((T_WARM, cnt), np.int32)
                for t in range(T_WARM):
                    bm.update(f=fs[t])
                    r = bm.solve()
                    gx[t], glam[t], gflag[t], git[t] = r["x"], r["lam"], r["exitflag"], r["iter"]
                bm.close()
                acc["gpu_s"] += time.perf_counter() - t0
                dt, x, lam, flag, it = O.timed_cpu_warm(ref, threads, q["H"], q["f"], q["A"], q["bupper"], q["blower"], fs, ms)
                acc["cpu_s"] += dt
            units = gflag.size
            acc["N"] += units
            acc["flag"] += int((gflag == flag).sum())
            acc["it"] += int((git == it).sum())
            acc["aset"] += int(np.all(np.sign(glam) == np.sign(lam), axis=-1).sum())
            acc["dx"] = max(acc["dx"], float(np.abs(gx - x).max()))
            acc["bits"] += int((np.all(np.ascontiguousarray(gx).view(np.uint64) == np.ascontiguousarray(x).view(np.uint64), axis=-1)
                                & np.all(np.ascontiguousarray(glam).view(np.uint64) == np.ascontiguousarray(lam).view(np.uint64), axis=-1)).sum())
            acc["iters"] += int(it.sum())
            acc["not_optimal"] += int((flag != 1).sum())
            print(f"  {cfg} [{lo}, {lo + cnt}): flags {acc['flag']}/{acc['N']} iters {acc['it']}/{acc['N']} active sets {acc['aset']}/{acc['N']} "
                  f"max|dx| {acc['dx']:.2e}", flush=True)
        unit = "warm solves" if cfg == "C5" else "QPs"
        rec = dict(shape=dict(n=n, m=m, ms=ms, n_active=na, seed=seed), units=acc["N"], unit=unit,
                   identical_exitflag=acc["flag"] / acc["N"], identical_iter=acc["it"] / acc["N"], identical_active_set=acc["aset"] / acc["N"],
                   bit_identical_x_and_lam=acc["bits"] / acc["N"], max_abs_dx=acc["dx"], mean_iterations=acc["iters"] / acc["N"], reference_not_optimal=acc["not_optimal"],
                   max_abs_x_minus_analytic_optimum=acc["dxref"] if cfg != "C5" else None,
                   gpu_wall_s_incl_pcie_and_python=round(acc["gpu_s"], 2), reference_cpu_s=round(acc["cpu_s"], 2),
                   reference_rate=f"{acc['N'] / acc['cpu_s']:.0f} {unit}/s on {threads} threads", generate_s=round(acc["gen_s"], 1))
        report["configs"][cfg] = rec
        print(f"{cfg}: {acc['N']} {unit}: flags {rec['identical_exitflag']:.6f} iters {rec['identical_iter']:.6f} active sets "
              f"{rec['identical_active_set']:.6f} bitwise x,lam {rec['bit_identical_x_and_lam']:.6f} max|dx| {rec['max_abs_dx']:.2e} | mean iterations {rec['mean_iterations']:.2f} | "
              f"reference {rec['reference_rate']}", flush=True)
    ok = all((not exact or r["bit_identical_x_and_lam"] == 1) and r["identical_exitflag"] == 1 and r["identical_iter"] == 1 and r["identical_active_set"] == 1 and r["max_abs_dx"] < 1e-9
             for r in report["configs"].values())
    report["all_identical"] = bool(ok)
    print(json.dumps(report))
    pool.close()
    # (leave through os._exit: INTEGRATION.md "Process exit")
    sys.stdout.flush()
    os._exit(0 if ok else 1)


if __name__ == "__main__":
    main()
