"""The image kernel (k_ldp_reg<NB, NP, true, IMG>: an fp32 image of M in the registers, two waves per SIMD) against the oracle and against
the full-register kernel: parity on C2 draws (cold solves, warm UPDATE_v / UPDATE_d sequences, forced hand-overs through DAQP_AMD_IMG_ROWS),
then the solve launch timed with and without it.   usage: python tools/img_probe.py [N_parity] [N_time] [rows,rows,...]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
import daqp_amd

NP_ = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
# "rows:cache" = DAQP_AMD_IMG_ROWS (working-set rows the image kernel holds; beyond: hand-over) : DAQP_AMD_IMG_CACHE (rows of them in LDS; beyond: scratch tier)
ROWS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["44:", "44:6", "30:10", "12:3"]


def set_rows(spec):
    r, _, c = spec.partition(":")
    os.environ["DAQP_AMD_IMG_ROWS"] = r
    if c:
        os.environ["DAQP_AMD_IMG_CACHE"] = c
    else:
        os.environ.pop("DAQP_AMD_IMG_CACHE", None)
os.environ["DAQP_AMD_NO_RECHECK"] = "1"
os.environ["DAQP_AMD_IMG_MIN_BATCH"] = "1"
n, m, ms, na, seed, _ = O.CONFIGS["C2"]
ora = O.Oracle()
q = O.generate_batch(NP_, n, m, ms, na, seed, start=500000)
ref = ora.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)


def compare(tag, g, r):
    same = (np.array_equal(g["exitflag"], r[3]), np.array_equal(g["iter"], r[4]), np.array_equal(np.sign(g["lam"]), np.sign(r[1])))
    dx = np.abs(g["x"] - r[0]).max()
    bad = np.nonzero((g["exitflag"] != r[3]) | (g["iter"] != r[4]))[0]
    print(f"{tag}: flags {same[0]} iter {same[1]} active sets {same[2]} max|dx| {dx:.2e}" + (f"  first bad {bad[:8]} gpu it {g['iter'][bad[:8]]} ref it {r[4][bad[:8]]} flags {g['exitflag'][bad[:8]]}" if len(bad) else ""), flush=True)
    return all(same) and dx < 1e-9


ok = True
for rows in ROWS:
    set_rows(rows)
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    ok &= compare(f"cold, image kernel, {rows} ", g, ref)
os.environ["DAQP_AMD_NO_IMG32"] = "1"
g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
ok &= compare("cold, full-register kernel", g, ref)
os.environ.pop("DAQP_AMD_NO_IMG32")

# warm sequences: UPDATE_v steps, then UPDATE_d steps, against the oracle's models
S = min(NP_, 1024)
rng = np.random.default_rng(7)
for rows in ROWS[:3]:
    set_rows(rows)
    bm = daqp_amd.BatchModel(S, n, m, ms)
    bm.setup(q["H"][:S], q["f"][:S], q["A"][:S], q["bupper"][:S], q["blower"][:S], None, init_mask=0)
    bm.solve()
    mods = []
    for k in range(S):
        md = ora.model(n, m, ms)
        md.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
        md.solve()
        mods.append(md)
    f = q["f"][:S].copy(); bu = q["bupper"][:S].copy(); bl = q["blower"][:S].copy()
    for t in range(6):
        if t % 2 == 0:
            f = f + 0.05 * rng.standard_normal(f.shape)
            bm.update(f=f)
        else:
            sh = 0.02 * rng.standard_normal(bu.shape)
            bu = bu + sh; bl = bl + sh
            bm.update(bupper=bu, blower=bl)
        g = bm.solve()
        rx = np.zeros((S, n)); rlam = np.zeros((S, m)); rfl = np.zeros(S, np.int32); rit = np.zeros(S, np.int32)
        for k, md in enumerate(mods):
            if t % 2 == 0:
                md.update(daqp_amd.UPDATE_v, f=f[k])
            else:
                md.update(daqp_amd.UPDATE_d, bupper=bu[k], blower=bl[k])
            r = md.solve()
            rx[k], rlam[k], rfl[k], rit[k] = r[0], r[1], r[3], r[4]
        ok &= compare(f"warm step {t} ({'f' if t % 2 == 0 else 'b'}), {rows} rows", g, (rx, rlam, None, rfl, rit))
    bm.close()
print("PARITY", "ALL OK" if ok else "FAILED", flush=True)

# timing: the solve launch of a cold step, then warm steps
import torch
from daqp_amd.synthetic import generate_batch_torch
qt = generate_batch_torch(NT, n, m, ms, na, seed=42, device="cuda:0")
mask = daqp_amd.UPDATE_unconstrained | daqp_amd.UPDATE_eliminate
for label, env in [(f"image kernel, {wv} workgroups per CU", {"DAQP_AMD_IMG_WAVES": str(wv)}) for wv in (5, 6, 7, 8)] + [("full-register kernel", {"DAQP_AMD_NO_IMG32": "1"})]:
    for k in ("DAQP_AMD_IMG_ROWS", "DAQP_AMD_NO_IMG32", "DAQP_AMD_IMG_CACHE", "DAQP_AMD_IMG_WAVES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    bm = daqp_amd.BatchModel(NT, n, m, ms, device=0)
    ts = []
    for it in range(6):
        bm.setup(qt["H"], qt["f"], qt["A"], qt["bupper"], qt["blower"], None, init_mask=mask)
        r = bm.solve(out="torch")
        torch.cuda.synchronize()
        ts.append(bm.kernel_ms())
    cold = np.median([b for a, b in ts[2:]]); setup = np.median([a for a, b in ts[2:]])
    bm.setup(qt["H"], qt["f"], qt["A"], qt["bupper"], qt["blower"], None, init_mask=0)
    bm.solve(out="torch")
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(45)
    cur = qt["f"]; wt = []
    for t in range(10):
        cur = cur + 0.05 * torch.randn(cur.shape, generator=gen, dtype=torch.float64, device="cuda:0")
        bm.update(f=cur)
        r = bm.solve(out="torch")
        torch.cuda.synchronize()
        wt.append(bm.kernel_ms()[1])
    print(f"{label}: setup {setup:.2f} ms, cold solve launch {cold:.2f} ms per {NT}, warm solve launch {np.median(wt):.2f} ms; it {float(r['iter'].double().mean()):.2f}, all optimal {bool((r['exitflag'] == 1).all())}", flush=True)
    bm.close()
