#!/usr/bin/env python
"""bench.py -- QPs/sec (fp64) for batched random dense QPs on 1/2/4/8 MI355X.

One "step" = the whole hot path (QP->LDP setup, dual active-set iteration, back-transform) over one batch of synthetic
QPs per GPU, inputs already resident in HBM, results left in HBM.  The headline (`value`) is BASELINE.json's metric:
config C2, 100 000 QPs per GPU, n=50 m=150.  The same JSON line carries the other BASELINE configs under "configs"
(C3: MPC-size QPs n=12 m=48 ms=12; C4: n=200 m=600; C5: warm-started sequence on the C2 batch), each with its own
throughput, solve-kernel roofline, CPU baseline sample and parity against that CPU run.

One process per GPU; independent QPs are sharded across ranks with no data-path collective; the only communication is
the barrier + MAX of the elapsed time the contract asks for.

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
           bench.py --gpus 8 --steps 20 --warmup 2                      # weak scaling: 100k C2 QPs per GPU
    ... bench.py --gpus 8 --config C3 --strong                          # ONE 1M-QP C3 batch, QP k on rank k mod 8
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
VALU_PEAK_GCYC = 1024 * 2.4  # 256 CUs x 4 SIMDs x 2.4 GHz: G SIMD-cycles/s in which a vector pipe can be issuing (same guide)
# what binds each configuration's solve launch (counters in profiles/r*_pmc_summary.json)
BOUND = {"C2": "issue", "C3": "issue", "C4": "hbm", "C5": "issue"}
BOUND_IS = {
    "issue": "instruction issue: M and the iterate are register-resident, HBM sees a few percent of the algorithmic bytes. achieved = VALU-busy "
             "SIMD-cycles of the launch (4 x SQ_ACTIVE_INST_VALU, committed counter pass of this library build) / launch time (HIP events, this run); "
             "peak = 1024 SIMDs x 2.4 GHz",
    "hbm": "the memory system: the scan (an fp32 image of M per iteration), the primal step and the Gram column (every active row, twice per iteration) "
           "stream through each CU's vector-memory path while all 256 CUs do the same.  achieved = HBM-side bytes of the launch (FETCH_SIZE x 2 + "
           "WRITE_SIZE of the committed counter pass of this library build) / launch time (HIP events, this run); peak = 8 TB/s.  The issue roof of the "
           "other configurations is quoted under `issue`",
}

# name -> (n, m, ms, active at the optimum, QPs per GPU (weak) / in total (strong), description)
CONFIGS = {
    "C2": dict(n=50, m=150, ms=0, na=20, per_gpu=100_000, total=100_000, what="random dense QPs"),
    "C3": dict(n=12, m=48, ms=12, na=6, per_gpu=125_000, total=1_000_000, what="MPC-size QPs with simple bounds (1M over 8 GPUs)"),
    "C4": dict(n=200, m=600, ms=0, na=80, per_gpu=10_000, total=10_000, what="large QPs (working set beyond LDS)"),
    "C5": dict(n=50, m=150, ms=0, na=20, per_gpu=100_000, total=100_000, what="warm-started sequence: T=10 steps f <- f + 0.05 N(0,I) on device-resident factors"),
}
METRIC = {
    "C2": "QPs/sec (fp64) for batched random dense QPs n=50 m=150",
    "C3": "QPs/sec (fp64) for batched MPC-size dense QPs n=12 m=48 ms=12",
    "C4": "QPs/sec (fp64) for batched random dense QPs n=200 m=600",
    "C5": "warm solves/sec (fp64) for a warm-started sequence of dense QPs n=50 m=150",
}


ARITH = {   # what "dtype: f64" means per configuration in the library's DEFAULT mode (the `exact` entry of each line is the other mode)
    "C2": "f64 throughout; fused multiply-adds in the iteration, M = A R^-1 on the f64 matrix cores (v_mfma_f64_16x16x4)",
    "C3": "f64 throughout; fused multiply-adds in the iteration; setup in the reference's operation order (16 problems per wavefront, no matrix cores at n = 12)",
    "C4": "f64 results; the feasibility scan is SCREENED in f32 under a rigorous error bound (the f64 scan decides whenever the f32 one is not certain: ~1 % of the scans), "
          "inverse factor W = L^-1 with tree sums, M = A R^-1 on the f64 matrix cores",
    "C5": "f64 throughout; fused multiply-adds in the iteration (factors set up once on the f64 matrix cores)",
}


def algorithmic_bytes(n, m, ms, iters, warm=False):
    """per-QP  B = B_io + it * 8 (m-ms) n; a warm step (C5) reads only the new f."""
    mA = m - ms
    b_in = 8 * n if warm else 8 * (n * n + n + mA * n + 2 * m) + 4 * m
    b_out = 8 * (n + m) + 8
    stream = iters.astype(np.float64) * 8.0 * mA * n
    return b_in, b_out, stream


CPU_LEG_S = 1.2     # every timed CPU leg lasts at least about this long (the sample is swept `passes` times inside the clock)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def timed_leg(run, units):
    """run(passes) -> (seconds, ...): one calibration call with one pass, then -- if that was shorter than CPU_LEG_S -- the
    measured call with as many passes as fill CPU_LEG_S.  Returns (units per second, wall seconds, passes, result of the call)."""
    r = run(1)
    if r[0] >= CPU_LEG_S:
        return units / r[0], r[0], 1, r
    passes = 1
    for _ in range(4):      # (the first, cold pass overstates the time per pass: rescale until the leg is long enough)
        passes = int(min(1 << 16, max(passes + 1, np.ceil(1.15 * passes * CPU_LEG_S / max(r[0], 1e-6)))))
        r = run(passes)
        if r[0] >= 0.85 * CPU_LEG_S:
            break
    return units * passes / r[0], r[0], passes, r


def cpu_baseline(q_host, ms, gpu_res, threads_options):
    """The same QPs solved one at a time by daqp_quadprog on host threads: the reference library itself (oracle/_ref, driven
    by oracle/ref_batch.c: one pthread per contiguous slice, threads created before the clock starts) when it travelled with
    the repo, else this repo's C restatement built with the reference's flags.  Every thread count of `threads_options` is
    tried, each for >= CPU_LEG_S seconds of wall time; reported: the best."""
    from oracle import oracle as O
    S = q_host["f"].shape[0]
    tried = ""
    kind = "reference" if O.reference_available() else "port"
    libpath = os.path.join(O.HERE, "_ref", "libdaqp_ref.so") if kind == "reference" else os.path.join(O.HERE, "liboracle_fast.so")
    best = None
    if kind == "reference":
        for th in threads_options:
            rate, wall, passes, r = timed_leg(lambda ps: O.timed_cpu_batch(libpath, th, q_host["H"], q_host["f"], q_host["A"], q_host["bupper"],
                                                                           q_host["blower"], ms, passes=ps), S)
            tried += f"{th} threads: {rate:.0f} QPs/s ({passes} passes, {wall:.2f} s); "
            if best is None or rate > best[1]:
                best = (th, rate, wall, passes, r)
        cores, rate, wall, passes, (_, x, lam, fval, flag, it) = best
        dt = S / rate
    else:
        import threading
        solver = O.Oracle(fast=True)
        n, m = q_host["f"].shape[1], q_host["bupper"].shape[1]
        x, lam = np.zeros((S, n)), np.zeros((S, m))
        flag, it = np.zeros(S, np.int32), np.zeros(S, np.int32)
        cores = threads_options[0]

        def work(lo, hi):   # one C call per slice; ctypes drops the GIL for its duration
            r = solver.quadprog_batch(q_host["H"][lo:hi], q_host["f"][lo:hi], q_host["A"][lo:hi],
                                      q_host["bupper"][lo:hi], q_host["blower"][lo:hi], None, ms=ms)
            x[lo:hi], lam[lo:hi], flag[lo:hi], it[lo:hi] = r[0], r[1], r[3], r[4]

        bounds = np.linspace(0, S, min(cores, S) + 1).astype(int)
        th = [threading.Thread(target=work, args=(bounds[i], bounds[i + 1])) for i in range(len(bounds) - 1)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        wall, passes = dt, 1
    parity = dict(
        sample=int(S),
        identical_active_set=float(np.mean(np.all(np.sign(lam) == np.sign(gpu_res["lam"]), axis=1))),
        identical_iter=float(np.mean(it == gpu_res["iter"])),
        identical_exitflag=float(np.mean(flag == gpu_res["exitflag"])),
        max_abs_dx=float(np.abs(x - gpu_res["x"]).max()),
    )
    return dict(value=S / dt, unit="QPs/s", cores=int(cores), kind=kind, cpu=cpu_model(), wall_s=wall, passes=passes,
                sample=f"first {S} QPs of the rank-0 batch swept {passes}x inside the clock, daqp_quadprog one QP at a time on {cores} host threads "
                       f"({S * 1.0 / dt / cores:.0f} QPs/s per thread), wall {wall:.2f} s" + (f" [best of: {tried.strip()}]" if tried else "")), parity, cores


def cpu_baseline_warm(q_host, fs_host, ms, thread_options):
    """C5 on the host cores, all in C (oracle/ref_batch.c::ref_warm_run): per QP the reference's setup_daqp -> daqp_solve
    (untimed), then T x {daqp_update_ldp(UPDATE_v) -> daqp_solve} over the first T steps of the walk of f, P pthreads with one
    contiguous slice each; only the warm phase is timed (first thread in to last thread out).  Best of the thread counts tried.
    Without oracle/_ref: the C restatement through ctypes on one thread (kind "port")."""
    from oracle import oracle as O
    T, S = fs_host.shape[0], q_host["f"].shape[0]
    n, m = q_host["f"].shape[1], q_host["bupper"].shape[1]
    if O.reference_available():
        libpath = os.path.join(O.HERE, "_ref", "libdaqp_ref.so")
        best, tried = None, ""
        for th in thread_options:
            rate, wall, passes, r = timed_leg(lambda ps: O.timed_cpu_warm(libpath, th, q_host["H"], q_host["f"], q_host["A"], q_host["bupper"],
                                                                          q_host["blower"], fs_host, ms, passes=ps), S * T)
            tried += f"{th} threads: {rate:.0f}/s ({passes} passes, {wall:.2f} s); "
            if best is None or rate > best[1]:
                best = (th, rate, wall, passes, r)
        cores, rate, wall, passes, (_, x, lam, flag, it) = best
        dt = S * T / rate
        kind = "reference"
        how = (f"C driver, {cores} pthreads created before the clock starts, no Python in the timed loop; the walk is swept {passes}x (forth and back: "
               f"every step one increment away from the previous one) [best of: {tried.strip()}]")
    else:
        drv = O.Oracle(fast=True)
        x, lam = np.zeros((T, S, n)), np.zeros((T, S, m))
        flag, it = np.zeros((T, S), np.int32), np.zeros((T, S), np.int32)
        dt = 0.0
        for k in range(S):
            md = drv.model(n, m, ms)
            md.setup(q_host["H"][k], q_host["f"][k], q_host["A"][k], q_host["bupper"][k], q_host["blower"][k], None)
            md.solve()
            t0 = time.perf_counter()
            for t in range(T):
                md.update(O.UPDATE_v, f=fs_host[t, k])
                r = md.solve()
                x[t, k], lam[t, k], flag[t, k], it[t, k] = r[0], r[1], r[3], r[4]
            dt += time.perf_counter() - t0
        cores, kind, how = 1, "port", "C restatement through ctypes on one thread (oracle/_ref did not travel)"
        wall, passes = dt, 1
    return dict(value=S * T / dt, unit="warm solves/s", cores=int(cores), kind=kind, cpu=cpu_model(), wall_s=wall, passes=passes,
                sample=f"first {S} QPs x the first {T} warm steps of the walk of f: daqp_update_ldp(UPDATE_v) + daqp_solve per step, "
                       f"setup_daqp + cold solve untimed; {how}; wall {wall:.2f} s"), dict(x=x, lam=lam, flag=flag, iter=it)


def warm_parity(bm_factory, q, fs, S, T, cpu):
    """the GPU's warm sequence replayed on the first S QPs from a fresh setup, compared with the CPU run at EVERY step:
    iteration count, exit flag, active set (index and side) and |dx|"""
    import daqp_amd
    bm = bm_factory(S)
    sl = {k: q[k][:S].contiguous() for k in ("H", "f", "A", "bupper", "blower")}
    bm.setup(sl["H"], sl["f"], sl["A"], sl["bupper"], sl["blower"], None, init_mask=0)
    bm.solve(out="torch")
    same_it = same_flag = same_as = 0.0
    dx = 0.0
    for t in range(T):
        bm.update(f=fs[t, :S].contiguous())
        g = bm.solve(out="torch")
        gi, gf = g["iter"].cpu().numpy(), g["exitflag"].cpu().numpy()
        same_it += float(np.mean(gi == cpu["iter"][t]))
        same_flag += float(np.mean(gf == cpu["flag"][t]))
        same_as += float(np.mean(np.all(np.sign(g["lam"].cpu().numpy()) == np.sign(cpu["lam"][t]), axis=1)))
        dx = max(dx, float(np.abs(g["x"].cpu().numpy() - cpu["x"][t]).max()))
    bm.close()
    return dict(sample=int(S), steps=int(T), identical_iter=same_it / T, identical_exitflag=same_flag / T,
                identical_active_set=same_as / T, max_abs_dx=dx,
                identical_iter_last_step=float(np.mean(gi == cpu["iter"][T - 1])), max_abs_dx_last_step=float(np.abs(g["x"].cpu().numpy() - cpu["x"][T - 1]).max()))


def transfer_times(torch, q, res, N, warm):
    """"H2D/D2H reported separately".  What a host caller pays on top of `value`: the step's inputs (B_io's arrays: H, f,
    A, bupper, blower -- a warm step: f only) from PINNED host memory to the device and its outputs (x, lam, exit flag, iter) back,
    timed with device events around the copies.  At most ~1 GiB per direction is pinned and copied (a slice of the batch, problems
    [0, S)); the whole batch's figure is that rate applied to all N problems -- both are in the record."""
    names = ("f",) if warm else ("H", "f", "A", "bupper", "blower")
    per_qp_in = sum(q[k][0].numel() * q[k].element_size() for k in names)
    outs = ("x", "lam", "exitflag", "iter")
    per_qp_out = sum(res[k][0].numel() * res[k].element_size() for k in outs)
    S = int(max(1, min(N, (1 << 30) // max(per_qp_in, 1))))
    host_in = [torch.empty(q[k][:S].shape, dtype=q[k].dtype).pin_memory() for k in names]
    dev_in = [torch.empty_like(q[k][:S]) for k in names]
    host_out = [torch.empty(res[k][:S].shape, dtype=res[k].dtype).pin_memory() for k in outs]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    best_in = best_out = 1e30
    for _ in range(3):
        torch.cuda.synchronize()
        e[0].record()
        for h, d in zip(host_in, dev_in):
            d.copy_(h, non_blocking=True)
        e[1].record()
        e[2].record()
        for k, h in zip(outs, host_out):
            h.copy_(res[k][:S], non_blocking=True)
        e[3].record()
        torch.cuda.synchronize()
        best_in, best_out = min(best_in, e[0].elapsed_time(e[1])), min(best_out, e[2].elapsed_time(e[3]))
    gbs_in, gbs_out = per_qp_in * S / best_in / 1e6, per_qp_out * S / best_out / 1e6
    return {"sample_qps": S, "h2d_bytes_per_qp": per_qp_in, "d2h_bytes_per_qp": per_qp_out, "h2d_GBps": gbs_in, "d2h_GBps": gbs_out,
            "h2d_ms": per_qp_in * N / gbs_in / 1e6, "d2h_ms": per_qp_out * N / gbs_out / 1e6,
            "what": "pinned host <-> device copies of the step's inputs / outputs (torch copy_, non_blocking, device events), measured on "
                    f"problems [0, {S}) and scaled to the batch of {N}; NOT part of `value` (inputs and outputs resident in HBM)"}


class Runner:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        assert self.world == args.gpus, "launch_plan() lets nothing else through"
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the solver has no CPU path")
        torch.cuda.set_device(self.local_rank)
        # under a torch.distributed launcher (RANK / MASTER_ADDR set) the process group is created even for ONE rank, so that
        # the RCCL communicator, the barrier and the MAX all-reduce on the device are the same code at every world size
        self.grouped = self.world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
        self.comm = None
        if self.grouped:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
                try:
                    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception:
                    ver = "?"
                self.comm = "nccl (RCCL %s)" % ver
            else:
                dist.init_process_group(args.backend)
                self.comm = args.backend
        self.red_device = "cuda" if (self.grouped and args.backend == "nccl") else "cpu"
        cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        # the fastest of {all, 1/2, 1/4, 1/8 of} the host threads: with every hardware thread busy the per-thread rate
        # collapses on these hosts (SMT + memory), and the baseline should be the CPU's best, not its most crowded
        self.thread_options = sorted({cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8)}, reverse=True)

    def sync(self):
        self.torch.cuda.synchronize()
        if self.grouped:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def generate(self, cfg, N_local, strong, N_total):
        """this rank's QPs: an independent shard (weak), or QP k of ONE batch of N_total on rank k mod world (strong: every
        rank draws the same global stream chunk by chunk and keeps its interleaved share, daqp_amd.parallel.shard_indices)"""
        from daqp_amd.synthetic import generate_batch_torch
        from daqp_amd.parallel import shard_indices
        torch = self.torch
        dev = f"cuda:{self.local_rank}"
        c = CONFIGS[cfg]
        if not strong:
            return generate_batch_torch(N_local, c["n"], c["m"], c["ms"], c["na"], seed=42 + 1000 * self.rank, device=dev)
        idx = shard_indices(N_total, self.rank, self.world)
        parts, chunk = [], 65536
        for s in range(0, N_total, chunk):
            B = min(chunk, N_total - s)
            g = generate_batch_torch(B, c["n"], c["m"], c["ms"], c["na"], seed=4242 + s, device=dev)
            mine = torch.from_numpy(idx[(idx >= s) & (idx < s + B)] - s).to(dev)
            parts.append({k: v[mine] for k, v in g.items()})
        return {k: torch.cat([p[k] for p in parts]) for k in parts[0]}

    def run(self, cfg, steps, warmup, batch=None, strong=False, cpu_sample=-1, thread_options=None, exact_steps=0):
        import daqp_amd
        from daqp_amd.parallel import max_over_ranks
        torch = self.torch
        c = CONFIGS[cfg]
        n, m, ms = c["n"], c["m"], c["ms"]
        N_total = (batch if batch else c["total"]) if strong else None
        N = len(range(self.rank, N_total, self.world)) if strong else (batch if batch else c["per_gpu"])
        q = self.generate(cfg, N, strong, N_total)
        bm = daqp_amd.BatchModel(N, n, m, ms, device=self.local_rank)
        mask = daqp_amd.UPDATE_unconstrained | daqp_amd.UPDATE_eliminate   # daqp_quadprog semantics
        warm = cfg == "C5"
        T = 10

        def cold_step():
            bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=mask)
            return bm.solve(out="torch")

        if warm:
            # T = 10 steps f <- f + 0.05 N(0,I) from the cold solve.  A pass = those T warm steps; further passes walk the SAME path
            # back (fs[T-2], ..., fs[0], the original f) and forth again -- every step one increment away from the previous one -- exactly as the CPU
            # leg does (oracle/ref_batch.c::ref_warm_run).  (Up to round 5 the walk went on for (warmup + steps) x T steps: after a hundred
            # increments f has drifted so far that 43 of 50 rows are active on average -- another workload than the one the survey names and the CPU
            # leg times; tools/c5_walk.py shows it.)
            gen = torch.Generator(device=q["f"].device)
            gen.manual_seed(45 + self.rank)
            fs = torch.empty((T,) + tuple(q["f"].shape), dtype=torch.float64, device=q["f"].device)
            cur = q["f"]
            for t in range(T):
                cur = cur + 0.05 * torch.randn(tuple(q["f"].shape), generator=gen, dtype=torch.float64, device=q["f"].device)
                fs[t] = cur
            f0 = q["f"].clone()
            cursor = [0]

            def step(record=None):
                res = None
                back = cursor[0] & 1
                cursor[0] += 1
                for j in range(T):
                    t = (T - 2 - j) if back else j
                    bm.update(f=fs[t] if t >= 0 else f0)
                    res = bm.solve(out="torch")
                    if record is not None:      # (untimed probe passes: every warm step's own launch time and iteration count)
                        record.append((bm.kernel_ms()[1], res["iter"].double().mean().item()))
                return res

            bm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=0)   # setup_daqp + the cold solve: untimed
            bm.solve(out="torch")
        else:
            step, fs = cold_step, None

        for _ in range(warmup):
            res = step()
        setup_ms, solve_ms = [], []
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
            if self.rank == 0:
                a, b = bm.kernel_ms()      # HIP events recorded on the launch stream around the kernels of the last launches
                setup_ms.append(0.0 if warm else a); solve_ms.append(b)
        self.sync()
        elapsed = time.perf_counter() - t0
        elapsed = max_over_ranks(elapsed, device=self.red_device)
        probe = None
        if warm and self.rank == 0:
            # the steps of a pass differ (working sets grow along the walk, the way back ends on the cold optimum): the launch time quoted for the
            # roofline is the mean over ONE forward and ONE backward pass, sampled step by step outside the timed region (a sample waits for the device)
            probe = []
            step(probe); step(probe)
            self.sync()
        exact = None
        if exact_steps > 0 and not warm and self.world == 1:
            # the same steps in the library's EXACT arithmetic (DAQP_AMD_EXACT=1: the reference's operation order throughout,
            # bit-identical results) on a second batch of workspaces -- reported next to the default mode's `value`
            os.environ["DAQP_AMD_EXACT"] = "1"
            try:
                bx = daqp_amd.BatchModel(N, n, m, ms, device=self.local_rank)
            finally:
                os.environ.pop("DAQP_AMD_EXACT", None)
            def xstep():
                bx.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, init_mask=mask)
                return bx.solve(out="torch")
            rx = xstep()
            self.sync()
            tx = time.perf_counter()
            for _ in range(exact_steps):
                rx = xstep()
            self.sync()
            dtx = time.perf_counter() - tx
            a, b_ = bx.kernel_ms()
            exact = dict(value=N * exact_steps / dtx, unit="QPs/s", steps=exact_steps, ms_per_step=dtx / exact_steps * 1e3, setup_ms=a, solve_ms=b_,
                         same_iterations_as_default=bool((rx["iter"] == res["iter"]).all().item()),
                         max_abs_dx_vs_default=float((rx["x"] - res["x"]).abs().max().item()),
                         what="DAQP_AMD_EXACT=1: every sum in the reference's operation order (no fused multiply-adds, no matrix cores, no fp32 screening, no inverse factor): results bit-identical to the reference")
            bx.close()
            del bx, rx
        rechecked = int(bm.rechecked())   # problems of the last step whose INFEASIBLE verdict took the second pass in the reference's arithmetic: none here
        return q, res, bm, dict(N=N, N_total=N_total, rechecked=rechecked, elapsed=elapsed, setup_ms=setup_ms, solve_ms=solve_ms, T=T if warm else 1, fs=fs, exact=exact, probe=probe)

    def report(self, cfg, q, res, info, steps, cpu_sample, headline):
        """rank 0: the JSON fields of one configuration"""
        c = CONFIGS[cfg]
        n, m, ms = c["n"], c["m"], c["ms"]
        warm = cfg == "C5"
        N, T = info["N"], info["T"]
        units = (info["N_total"] if info["N_total"] else self.world * N) * steps * T
        iters = res["iter"].cpu().numpy()
        if warm and info.get("probe"):      # mean over the warm steps of a forward and a backward pass (run.probe), not the last step's
            iters = np.full(iters.shape, float(np.mean([p_[1] for p_ in info["probe"]])))
        b_in, b_out, stream = algorithmic_bytes(n, m, ms, iters, warm)
        ldp_bytes = float(stream.sum() + b_out * N)           # what one solve launch must move (SURVEY 8d without the inputs)
        all_bytes = float(stream.sum() + (b_in + b_out) * N)  # SURVEY 8(d) B summed over the batch
        io_bytes = float((b_in + b_out) * N)                  # B_io only: the floor if M never leaves the chip
        t_ldp = float(np.mean([p_[0] for p_ in info["probe"]]) if (warm and info.get("probe")) else np.mean(info["solve_ms"])) * 1e-3
        t_setup = float(np.mean(info["setup_ms"])) * 1e-3
        ach = ldp_bytes / t_ldp / 1e9
        flags_ok = bool((res["exitflag"] == 1).all().item())
        hbm_eff = {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ldp_bytes,
                   "is": "EFFECTIVE rate: SURVEY 8(d)'s algorithmic bytes (M streamed once per iteration) over the launch time -- M is held on chip "
                         "(C4: screened through an fp32 image), so this says how fast the nominal traffic is served and may exceed 1; it is NOT the "
                         "fraction of a roof.  The HBM side's real utilisation is traffic_frac"}
        roof = {"bound": BOUND[cfg], "bound_is": BOUND_IS[BOUND[cfg]],
                "kernel": "solve launch (dual active-set iteration + back-transform" + (", fused UPDATE_v" if warm else "") + ")",
                "achieved": None, "peak": HBM_PEAK_GBS if BOUND[cfg] == "hbm" else VALU_PEAK_GCYC,
                "unit": "GB/s" if BOUND[cfg] == "hbm" else "G VALU-busy SIMD-cycles/s", "frac": None,
                "avg_launch_ms": t_ldp * 1e3, "traffic": None, "hbm_effective": hbm_eff,
                # the step time against B_io alone (inputs + outputs of the step): what is left of an HBM bound if M stays on chip
                "floor_frac": io_bytes / max(t_ldp + t_setup, 1e-12) / 1e9 / HBM_PEAK_GBS}
        out = {
            "metric": METRIC[cfg], "value": units / info["elapsed"], "unit": "warm solves/s" if warm else "QPs/s",
            "ms_per_step": info["elapsed"] / steps * 1e3, "steps": steps,
            "workload": f"{cfg}: {N} {c['what']} per GPU" + (f" (ONE batch of {info['N_total']}, QP k on rank k mod {self.world})" if info["N_total"] else "")
                        + f", n={n} m={m} ms={ms}, {c['na']} active at the optimum, kappa=100 (reference generate_test_QP restated in torch on the GPU, "
                        + "daqp_amd/synthetic.py: same family as SURVEY 8d's numpy default_rng([seed, k]) stream, not the same draws), "
                        + ("setup_daqp + cold solve untimed, then per step T=10 x {daqp_update_ldp(UPDATE_v) + daqp_solve} along the first 10 steps of the walk of f, forth and back (the CPU leg's walk)" if warm else
                           "daqp_quadprog semantics: setup + solve per step") + ", inputs and outputs resident in HBM",
            "batch_per_gpu": N, "mean_iterations": float(iters.mean()),
            "roofline": roof,
            "checks": {"all_optimal": flags_ok, "rechecked_in_exact_arithmetic": info["rechecked"]},
            "arith": ARITH[cfg],
        }
        if info.get("exact"):
            out["exact"] = info["exact"]
        if self.world == 1:
            try:
                out["transfers"] = transfer_times(self.torch, q if not warm else dict(q, f=info["fs"][0]), res, N, warm)
                tr = out["transfers"]
                out["transfers"]["value_with_transfers"] = units / (info["elapsed"] + steps * T * (tr["h2d_ms"] + tr["d2h_ms"]) * 1e-3)
            except Exception as ex:     # (not enough pinnable host memory on this box: say so, do not fail the bench)
                out["transfers"] = {"error": repr(ex)[:200]}
        if not warm:
            roof["pipeline"] = {"hbm_effective": all_bytes / (t_ldp + t_setup) / 1e9, "hbm_effective_frac": all_bytes / (t_ldp + t_setup) / 1e9 / HBM_PEAK_GBS,
                                "setup_ms": t_setup * 1e3, "solve_ms": t_ldp * 1e3, "algorithmic_bytes_per_step": all_bytes}
            out["checks"]["max_abs_x_minus_analytic_optimum"] = float((res["x"] - q["xref"]).abs().max().item())
        prof = committed_counters(cfg, N)
        roof["library"] = library_stamp()
        if prof.get("stale"):
            # no committed counter pass was taken with THIS build of the library (or with this batch size): nothing is quoted
            roof["stale"] = True
            roof["stale_why"] = prof["why"]
        else:
            roof["traffic"] = prof.get("traffic")
            roof["traffic_source"] = prof["traffic_source"]
            for k in ("binding", "issue"):
                if k in prof:
                    roof[k] = prof[k]
            if prof.get("traffic"):     # what the launch really moves through the memory side, against the peak: the HBM side's utilisation
                roof["traffic_frac"] = prof["traffic"] / max(t_ldp, 1e-12) / 1e9 / HBM_PEAK_GBS
            if prof.get("issue"):       # achieved / attainable on the issue roof, with the launch time measured in THIS run
                att = prof["issue"]["attainable_ms"]
                roof["issue"]["achieved_ms"] = t_ldp * 1e3
                roof["issue"]["frac"] = att / max(t_ldp * 1e3, 1e-12)
                # VALU-busy cycles of the launch (counter pass) / launch time (this run) against 1024 SIMDs x 2.4 GHz
                if BOUND[cfg] == "issue":
                    roof["achieved"] = att * 1e-3 * VALU_PEAK_GCYC / max(t_ldp, 1e-12)
                    roof["frac"] = roof["achieved"] / VALU_PEAK_GCYC
            if prof.get("setup") and not warm and "hbm_read_bytes" in prof["setup"] and not prof.get("setup_launches"):
                # the setup launch (QP -> LDP) on the HBM roof: the bytes it moved through the memory side (counter pass of this build:
                # FETCH_SIZE x 2 + WRITE_SIZE) over its duration (HIP events, this run); next to it what it MUST move -- the inputs once in,
                # the LDP once out -- and the issue-side counters that say what it waits for
                st = prof["setup"]
                moved = st["hbm_read_bytes"] + st["hbm_written_bytes"]
                must = float((b_in + 8 * (m * n + n * (n + 1) // 2 + n + 3 * m) + 4 * m) * N)
                roof["setup"] = {"kernel": st.get("kernel"), "bound": "hbm", "avg_launch_ms": t_setup * 1e3, "traffic": moved,
                                 "achieved": moved / max(t_setup, 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": moved / max(t_setup, 1e-12) / 1e9 / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": must, "algorithmic_frac": must / max(t_setup, 1e-12) / 1e9 / HBM_PEAK_GBS,
                                 "read_bytes": st["hbm_read_bytes"], "written_bytes": st["hbm_written_bytes"], "issue": st.get("issue")}
            if prof.get("setup_launches") and not warm:
                # a setup of several launches (C4: factorisation, general rows, the rest): ONE record per kernel -- that kernel's counter bytes over
                # that kernel's own average duration (both from the committed round profile: rocprofv3 --pmc passes and --kernel-trace --stats of the
                # same build; this run's HIP events only see the whole setup: `setup_ms_this_run`), and for the two matrix-core kernels the
                # algorithmic flops against the fp64 matrix peak (MI355X_MICROARCH.md: 78.6 TFLOP/s, the vector pipe's rate)
                F64_PEAK_TF = 78.6
                flops = {"k_fact_wg": 2.0 * n ** 3 / 3.0 * N, "k_setup_m": 1.0 * (m - ms) * n * n * N}      # Cholesky + inverse; M = A R^-1 against a triangular R^-1
                recs = []
                for st in prof["setup_launches"]:
                    ms_k = st.get("avg_ms_kernel_trace")
                    moved = st.get("hbm_read_bytes", 0.0) + st.get("hbm_written_bytes", 0.0)
                    rec = {"kernel": st["kernel"], "bound": "hbm", "avg_launch_ms": ms_k, "avg_launch_ms_source": "kernel trace of the committed round profile",
                           "traffic": moved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "read_bytes": st.get("hbm_read_bytes"), "written_bytes": st.get("hbm_written_bytes"),
                           "issue": st.get("issue")}
                    if ms_k:
                        rec["achieved"] = moved / (ms_k * 1e-3) / 1e9
                        rec["frac"] = rec["achieved"] / HBM_PEAK_GBS
                        fl = next((v for k_, v in flops.items() if k_ in st["kernel"]), None)
                        if fl:
                            rec["flops"] = fl
                            rec["tflops"] = fl / (ms_k * 1e-3) / 1e12
                            rec["mfma_frac"] = rec["tflops"] / F64_PEAK_TF
                    recs.append(rec)
                roof["setup"] = recs
                roof["setup_ms_this_run"] = t_setup * 1e3
            if BOUND[cfg] == "hbm" and prof.get("traffic"):     # the memory system is the roof: measured HBM-side bytes over the launch time
                roof["achieved"] = prof["traffic"] / max(t_ldp, 1e-12) / 1e9
                roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
                pat = pattern_ceiling()
                if pat:     # what the memory system delivers to THIS access pattern with every CU streaming (tools/ubench5.hip, committed run)
                    roof["pattern"] = dict(pat, frac=roof["achieved"] / pat["peak"])
        if cpu_sample > 0 and self.world == 1:
            S = min(N, cpu_sample)
            if warm:
                import daqp_amd
                qh = {k: q[k][:S].cpu().numpy() for k in ("H", "f", "A", "bupper", "blower")}
                base, cpu = cpu_baseline_warm(qh, info["fs"][:T, :S].cpu().numpy(), ms, self.thread_options)
                parity = warm_parity(lambda S_: daqp_amd.BatchModel(S_, n, m, ms, device=self.local_rank), q, info["fs"], S, T, cpu)
            else:
                qh = {k: q[k][:S].cpu().numpy() for k in ("H", "f", "A", "bupper", "blower")}
                gh = {k: res[k][:S].cpu().numpy() for k in ("x", "lam", "iter", "exitflag")}
                base, parity, best = cpu_baseline(qh, ms, gh, self.thread_options)
            out["cpu_baseline"] = base
            out["parity_vs_cpu"] = parity
        return out


def pattern_ceiling():
    """the chip-wide rate of tools/ubench5.hip (the C4 solve kernel's memory phases as a pure streaming pattern: 512 workgroups of 4 waves -- 256 of 8 in runs before round 6's tiered launch --,
    each re-reading its own 0.8 MB working set with 16-byte loads, 16 in flight per lane) from the newest committed run"""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ubench5.txt")))
    if not files:
        return None
    text = open(files[-1]).read()
    # (round 6: C4's cold solves run two four-wave workgroups per CU -- the ceiling is taken at THAT residency when the run has it)
    rates = [float(m.group(1)) for m in re.finditer(r"^512 workgroups x 4 waves, 1 working set.*?-> ([0-9.]+) TB/s", text, re.M)]
    if not rates:
        rates = [float(m.group(1)) for m in re.finditer(r"^256 workgroups x 8 waves, 1 working set.*?-> ([0-9.]+) TB/s", text, re.M)]
    if not rates:
        return None
    return {"peak": 1e3 * float(np.median(rates)), "unit": "GB/s", "source": "profiles/" + os.path.basename(files[-1]),
            "is": "what the memory system delivers to the kernel's own access pattern with all 256 CUs streaming (cache-resident or not: the same); "
                  "the launch's HBM-side bytes over its time against THAT"}


def library_stamp():
    """what identifies the build of the library the counters belong to: its version string and a hash of every file under
    daqp_amd/csrc (the same function stamps profiles/r*_pmc_summary.json, tools/pmc_config_summary.py)"""
    import glob
    import hashlib
    import daqp_amd
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "daqp_amd", "csrc", "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode() + b"\0")
            with open(f, "rb") as fh:
                h.update(fh.read())
    return {"version": daqp_amd.lib().daqp_amd_version().decode(), "csrc_sha16": h.hexdigest()[:16]}


def committed_counters(cfg, N):
    """HBM bytes per solve launch and the issue-side counters from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE
    and the SQ groups each in its own run, corrected as MI355X_MICROARCH.md prescribes; profiles/README.md).  Counters cannot be
    collected from inside this process, so the figures are those measured with this same workload -- and they are quoted ONLY
    from a summary stamped with the build of the library that is loaded now (library_stamp) and taken at this batch size;
    otherwise the record says `stale` and carries no counter-derived number."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return {"stale": True, "why": "no profiles/r*_pmc_summary.json"}
    now = library_stamp()
    why = "no committed counter summary carries this configuration"
    for f in reversed(files):
        doc = json.load(open(f))
        d = doc.get(cfg)
        if not d:
            continue
        if doc.get("_stamp") != now:
            why = f"newest summary with {cfg} ({os.path.basename(f)}) was taken with library {doc.get('_stamp')}, loaded is {now}"
            break
        if d.get("batch") != N:
            why = f"{os.path.basename(f)} holds {cfg} at batch {d.get('batch')}, this run uses {N}"
            break
        out = {"traffic": d.get("traffic_bytes_per_launch"), "traffic_source": "profiles/" + os.path.basename(f)}
        if "binding" in d:
            out["binding"] = d["binding"]
        if "issue" in d:
            out["issue"] = dict(d["issue"])
        if d.get("setup"):
            out["setup"] = dict(d["setup"], kernel=d.get("setup_kernel"))
        if d.get("setup_launches"):
            out["setup_launches"] = d["setup_launches"]
        return out
    return {"stale": True, "why": why}


def _r(v, sig=6):
    """floats to `sig` significant digits (the compact line); everything else unchanged"""
    if isinstance(v, float):
        return float(f"{v:.{sig}g}")
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def _short_roofline(r, with_setup=True):
    """the contract keys of a roofline record -- numbers only, no prose"""
    out = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "traffic", "traffic_source")}
    out["kernel"] = (r.get("kernel_name") or r.get("kernel") or "")[:80]
    he = r.get("hbm_effective") or {}
    out["hbm_effective"] = {k: he.get(k) for k in ("achieved", "peak", "unit", "frac") if k in he}
    for k in ("traffic_frac", "floor_frac"):
        if k in r:
            out[k] = r[k]
    if r.get("stale"):
        out["stale"] = True
    if r.get("issue"):
        out["issue"] = {k: r["issue"].get(k) for k in ("attainable_ms", "one_wave_per_simd_floor_ms", "achieved_ms", "frac") if k in r["issue"]}
    if r.get("pattern"):
        out["pattern"] = {k: r["pattern"].get(k) for k in ("peak", "unit", "frac")}
    if with_setup and r.get("setup"):
        st = r["setup"] if isinstance(r["setup"], list) else [r["setup"]]
        out["setup"] = [{k: e.get(k) for k in ("kernel", "bound", "avg_launch_ms", "traffic", "achieved", "peak", "unit", "frac", "tflops", "mfma_frac") if e.get(k) is not None}
                        for e in st]
    if r.get("pipeline"):
        out["setup_ms"], out["solve_ms"] = r["pipeline"].get("setup_ms"), r["pipeline"].get("solve_ms")
    return out


def _short_cpu(c):
    if not c:
        return None
    out = {k: c.get(k) for k in ("value", "unit", "cores", "kind", "cpu", "wall_s")}
    out["sample"] = str(c.get("sample", "")).split(" [best of")[0][:160]
    return out


def _short_parity(p_):
    if not p_:
        return None
    return {k: p_[k] for k in ("sample", "steps", "identical_active_set", "identical_iter", "identical_exitflag", "max_abs_dx") if k in p_}


def compact(line, full_path):
    """The ONE line the driver parses: the contract keys only, numbers without prose, <= 6 KB (a driver that keeps the last 8 KB of
    stdout must see the whole object -- round 5's 22 KB line did not parse).  Everything else (notes, transfers, batch sweep, exact-mode
    figures, issue counters, full workload descriptions, per-thread-count CPU legs) is in the full record written to `full_path`."""
    cfg = line["config"]
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": cfg["workload"].split(", kappa")[0][:140], "batch_per_gpu": cfg["batch_per_gpu"],
                     "mean_iterations": cfg["mean_iterations"], "parallelism": cfg["parallelism"].split(",")[0][:60] + ", shards only, no data-path collective"}
    if "entry" in cfg:
        out["config"]["entry"] = cfg["entry"]
    out["roofline"] = _short_roofline(line["roofline"])
    if line.get("cpu_baseline"):
        out["cpu_baseline"] = _short_cpu(line["cpu_baseline"])
    if line.get("parity_vs_cpu"):
        out["parity_vs_cpu"] = _short_parity(line["parity_vs_cpu"])
    if line.get("checks"):
        out["checks"] = line["checks"]
    if line.get("configs"):
        out["configs"] = {}
        for name, s_ in line["configs"].items():
            r = s_["roofline"]
            e = {"value": s_["value"], "unit": s_["unit"], "ms_per_step": s_["ms_per_step"], "batch_per_gpu": s_.get("batch_per_gpu"),
                 "mean_iterations": s_.get("mean_iterations"),
                 "roofline": {"bound": r.get("bound"), "frac": r.get("frac"), "avg_launch_ms": r.get("avg_launch_ms"), "traffic": r.get("traffic"),
                              "hbm_effective_frac": (r.get("hbm_effective") or {}).get("frac")}}
            if r.get("pattern"):
                e["roofline"]["pattern_frac"] = r["pattern"].get("frac")
            if s_.get("cpu_baseline"):
                e["cpu_baseline"] = {k: s_["cpu_baseline"].get(k) for k in ("value", "cores", "kind")}
            if s_.get("parity_vs_cpu"):
                e["parity_vs_cpu"] = {k: s_["parity_vs_cpu"].get(k) for k in ("sample", "identical_active_set", "identical_iter", "max_abs_dx")}
            if s_.get("checks"):
                e["all_optimal"] = s_["checks"].get("all_optimal")
            out["configs"][name] = e
    out["full"] = os.path.relpath(full_path, ROOT) if full_path else None
    return _r(out)


def emit(line, args):
    """rank 0: the full record to --full-out (default bench_full.json next to this file; a copy under gpurun_out/ when that directory
    exists, so that it comes back from a gpurun call), then the compact line as the LAST line of stdout"""
    path = args.full_out or os.path.join(ROOT, "bench_full.json")
    try:
        with open(path, "w") as fh:
            json.dump(line, fh, indent=1)
        side = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(side) and not args.full_out:
            with open(os.path.join(side, "bench_full.json"), "w") as fh:
                json.dump(line, fh, indent=1)
    except OSError:
        path = None
    text = json.dumps(compact(line, path), separators=(",", ":"))
    assert len(text) <= 6144, f"compact bench line is {len(text)} bytes"
    sys.stdout.flush()
    print(text, flush=True)


def launch_plan(gpus, env, argv):
    """What `bench.py --gpus N` has to do about ranks.  Under a torch.distributed launcher (WORLD_SIZE set) the launcher's world
    size must BE --gpus (anything else is a mis-launch and fails loudly); without one, N == 1 runs in this process (returns
    None) and N > 1 returns the command that re-executes this script as N ranks under torch.distributed.run."""
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but the launcher started WORLD_SIZE={ws} ranks; start as many ranks as --gpus says")
        return None
    if gpus == 1:
        return None
    import socket
    with socket.socket() as sk:           # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(cmd, args):
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not args.single_device and have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {have} HIP device(s) are visible (one rank per GPU; --single-device puts every rank on cuda:0 for tests)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def multi_entry(args):
    """--multi-entry: the single-process path over G devices -- daqp_batch_create_multi / setup_multi_shards / solve_multi_shards, problem k
    of the batch on shard k mod G, every shard's inputs and results resident on ITS device -- timed like the rank-per-GPU path (W warm-up
    steps, K timed steps between device-wide synchronisations) and reported in the same line format (scaling "weak": per-shard work fixed).
    No process group: the library's own host threads drive the shards."""
    import torch
    import daqp_amd
    from daqp_amd.synthetic import generate_batch_torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the solver has no CPU path")
    G = args.gpus
    have = torch.cuda.device_count()
    if not args.single_device and have < G:
        raise SystemExit(f"bench.py: --gpus {G} --multi-entry but {have} HIP device(s) are visible (--single-device lists device 0 {G} times)")
    devices = [0] * G if args.single_device else list(range(G))
    cfg = args.config
    if cfg == "C5":
        raise SystemExit("bench.py --multi-entry: C2, C3, C4 (cold steps)")
    c = CONFIGS[cfg]
    n, m, ms = c["n"], c["m"], c["ms"]
    per = args.batch or c["per_gpu"]
    N = per * G
    mb = daqp_amd.MultiBatchModel(N, n, m, ms, 0, devices=devices)
    shards = []
    for g in range(mb.shards):
        _, sn, dev = mb.shard(g)
        shards.append(generate_batch_torch(sn, n, m, ms, c["na"], seed=42 + 1000 * g, device=f"cuda:{dev}"))
    mask = daqp_amd.UPDATE_unconstrained | daqp_amd.UPDATE_eliminate

    def sync():
        for dev in sorted(set(devices)):
            torch.cuda.synchronize(dev)

    def step():
        mb.setup_shards(shards, init_mask=mask)
        return mb.solve_shards(out="torch")

    for _ in range(args.warmup):
        res = step()
    sync()
    setup_ms, solve_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        a, b = mb.shard_kernel_ms(0)
        setup_ms.append(a); solve_ms.append(b)
    sync()
    elapsed = time.perf_counter() - t0
    iters = np.concatenate([r["iter"].cpu().numpy() for r in res])
    b_in, b_out, stream = algorithmic_bytes(n, m, ms, iters[: shards[0]["f"].shape[0]], False)
    t_ldp = float(np.mean(solve_ms)) * 1e-3
    ldp_bytes = float(stream.sum() + b_out * shards[0]["f"].shape[0])
    ok = all(bool((r["exitflag"] == 1).all().item()) for r in res)
    dx = max(float((r["x"] - s_["xref"]).abs().max().item()) for r, s_ in zip(res, shards))
    line = {"metric": METRIC[cfg], "value": N * args.steps / elapsed, "unit": "QPs/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{cfg}: {per} {c['what']} per shard, n={n} m={m} ms={ms} (generator of daqp_amd/synthetic.py), daqp_quadprog semantics: setup + solve per "
                                   "step, every shard's inputs and outputs resident in its device's HBM",
                       "batch_per_gpu": per, "mean_iterations": float(iters.mean()), "entry": "multi",
                       "parallelism": f"ONE process, {mb.shards} shard(s) on device(s) {devices} through daqp_batch_setup_multi_shards / daqp_batch_solve_multi_shards "
                                      "(a host thread and a stream per shard, problem k on shard k mod G, no exchange step)"},
            "roofline": {"bound": BOUND[cfg], "kernel": "solve launch of shard 0", "avg_launch_ms": t_ldp * 1e3, "setup_ms_shard0": float(np.mean(setup_ms)),
                         "hbm_effective": {"achieved": ldp_bytes / max(t_ldp, 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": ldp_bytes / max(t_ldp, 1e-12) / 1e9 / HBM_PEAK_GBS},
                         "achieved": None, "peak": HBM_PEAK_GBS if BOUND[cfg] == "hbm" else VALU_PEAK_GCYC, "frac": None, "traffic": None,
                         "note": "counter-derived figures are quoted by the rank-per-GPU line (same kernels, same batch per device)"},
            "checks": {"all_optimal": ok, "max_abs_x_minus_analytic_optimum": dx}}
    mb.close()
    emit(line, args)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS), help="the configuration `value` is quoted on (BASELINE headline: C2)")
    ap.add_argument("--batch", type=int, default=0, help="QPs per GPU per step (0: the config's own; with --strong: in total)")
    ap.add_argument("--strong", action="store_true", help="ONE batch split over the ranks (QP k -> rank k mod G) instead of one batch per rank")
    ap.add_argument("--side-configs", default="auto", help="comma list of further configs reported under \"configs\" (auto: C3,C4,C5 at N=1 for the C2 headline; none)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="QPs for the CPU baseline (-1: auto, 0: skip)")
    ap.add_argument("--single-device", action="store_true", help="every rank uses cuda:0 (multi-rank smoke test on a 1-GPU box; use --backend gloo)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch-size sweep of the headline configuration (reported under \"batch_sweep\")")
    ap.add_argument("--multi-entry", action="store_true", help="ONE process drives --gpus devices through the C ABI's persistent multi-device batch "
                    "(daqp_batch_*_multi_shards: a host thread + stream per shard, inputs resident on each shard's device) instead of one rank per GPU; "
                    "with --single-device every shard sits on device 0")
    ap.add_argument("--full-out", default="", help="where the FULL record goes (default: bench_full.json next to bench.py); stdout carries the compact contract line only")
    ap.add_argument("--no-exact", action="store_true", help="skip the extra steps in the library's exact arithmetic mode (reported under \"exact\")")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.multi_entry:
        return multi_entry(args)
    plan = launch_plan(args.gpus, os.environ, sys.argv[1:])
    if plan is not None:
        # plain `python bench.py --gpus N` (no launcher): start the N ranks here, rank r on device r, and hand back their exit
        # code; rank 0's JSON line goes to this process's stdout through the launcher
        sys.exit(self_launch(plan, args))

    R = Runner(args)
    auto_sample = {"C2": 65536, "C3": 262144, "C4": 1024, "C5": 8192}   # ~10-25 CPU-seconds each on one core-group
    sample = lambda cfg: 0 if (args.cpu_sample == 0 or R.world > 1) else (auto_sample[cfg] if args.cpu_sample < 0 else args.cpu_sample)

    exact_steps = {"C2": 3, "C3": 5, "C4": 1, "C5": 0}
    want_exact = (not args.no_exact) and R.world == 1
    q, res, bm, info = R.run(args.config, args.steps, args.warmup, batch=args.batch or None, strong=args.strong,
                             exact_steps=exact_steps[args.config] if want_exact else 0)
    line = None
    if R.rank == 0:
        h = R.report(args.config, q, res, info, args.steps, sample(args.config), headline=True)
        line = {
            "metric": h["metric"], "value": h["value"], "unit": h["unit"], "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": h["ms_per_step"], "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": h["workload"], "batch_per_gpu": h["batch_per_gpu"], "mean_iterations": h["mean_iterations"],
                       "parallelism": f"{R.world} rank(s), one per GPU, independent shards (QP k -> rank k mod {R.world} with --strong), no collective in the data path"
                                      + (f"; barrier + MAX(elapsed) over {R.comm}, {R.world} rank(s)" if R.comm else "")},
            "roofline": h["roofline"], "checks": h["checks"], "arith": h["arith"],
        }
        if "exact" in h:
            line["exact"] = h["exact"]
        for k in ("transfers", "cpu_baseline", "parity_vs_cpu"):
            if k in h:
                line[k] = h[k]
    bm.close()
    del q, res, bm
    R.torch.cuda.empty_cache()

    if R.world == 1 and args.config == "C2" and not args.batch and not args.strong and not args.no_sweep:
        # where does the device saturate at one wave per SIMD?  The headline configuration at smaller batches (same generator, a few steps)
        sweep = []
        for nb in (1000, 10000):
            q, res, bm, info = R.run("C2", 5, 2, batch=nb)
            sweep.append({"batch": nb, "value": nb * 5 / info["elapsed"], "unit": "QPs/s", "ms_per_step": info["elapsed"] / 5 * 1e3,
                          "setup_ms": float(np.mean(info["setup_ms"])), "solve_ms": float(np.mean(info["solve_ms"]))})
            bm.close()
            del q, res, bm
            R.torch.cuda.empty_cache()
        if R.rank == 0:
            sweep.append({"batch": line["config"]["batch_per_gpu"], "value": line["value"], "unit": "QPs/s", "ms_per_step": line["ms_per_step"],
                          "setup_ms": line["roofline"].get("pipeline", {}).get("setup_ms"), "solve_ms": line["roofline"].get("pipeline", {}).get("solve_ms")})
            line["batch_sweep"] = sweep

    side = args.side_configs
    if side == "auto":
        side = "C3,C4,C5" if (R.world == 1 and args.config == "C2" and not args.batch and not args.strong) else "none"
    side_steps = {"C3": (10, 2), "C4": (3, 1), "C5": (10, 2), "C2": (5, 1)}
    if side != "none":
        cfgs = {}
        for cfg in [s for s in side.split(",") if s and s != args.config]:
            st, wu = side_steps[cfg]
            q, res, bm, info = R.run(cfg, st, wu, exact_steps=exact_steps[cfg] if want_exact else 0)
            if R.rank == 0:
                r = R.report(cfg, q, res, info, st, sample(cfg), headline=False)
                cfgs[cfg] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "workload", "batch_per_gpu", "mean_iterations",
                                               "roofline", "checks", "arith", "exact", "transfers", "cpu_baseline", "parity_vs_cpu") if k in r}
            bm.close()
            del q, res, bm
            R.torch.cuda.empty_cache()
        if R.rank == 0:
            line["configs"] = cfgs
    if R.rank == 0:
        emit(line, args)
    if R.grouped:
        R.dist.barrier()
        R.dist.destroy_process_group()


if __name__ == "__main__":
    main()
