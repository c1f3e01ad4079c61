"""How often would a reduced-precision screening image of M leave the feasibility scan undecided on config C2?  CPU only: the oracle's event
trace gives every working set a scan ran on; u is recomputed for it (minimum-norm solution on the active rows), the exact slacks of all open
rows are formed, and for an error bound E(precision) |u| the rows within 2E of the most violated one (or within E of their threshold on the
final scan) are counted -- those are the rows a screening pass would have to re-evaluate in fp64.   usage: python tools/band_stats.py [N]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, m, ms, na, seed, _ = O.CONFIGS["C2"]
ora = O.Oracle()
q = O.generate_batch(N, n, m, ms, na, seed)
PREC = {"fp32 (n/4+8) 2^-24": (n / 4 + 8) * 2.0 ** -24, "fp16 image + fp16 u (2 x 2^-11 + n 2^-24)": 2 * 2.0 ** -11 + n * 2.0 ** -24,
        "fp16 image + fp32 u (2^-11 + n 2^-24)": 2.0 ** -11 + n * 2.0 ** -24, "bf16 image + fp32 u (2^-8)": 2.0 ** -8 + n * 2.0 ** -24}
stat = {k: dict(scans=0, undecided=0, cand=0, final=0, final_band=0, final_rows=0) for k in PREC}
tol = 1e-6
for k in range(N):
    md = ora.model(n, m, ms); md.enable_trace(8000)
    md.setup(q["H"][k], q["f"][k], q["A"][k], q["bupper"][k], q["blower"][k], None)
    md.solve()
    M, R, v, du, dl, sc = md.ldp()
    W = []           # (row, side)
    def u_of(W):
        if not W: return np.zeros(n)
        MA = M[[r for r, _ in W]]; dA = np.array([du[r] if s > 0 else dl[r] for r, s in W])
        return MA.T @ np.linalg.lstsq(MA @ MA.T, dA, rcond=None)[0]
    def scan(W, final):
        u = u_of(W); nu = np.linalg.norm(u)
        mu = M @ u
        s = np.minimum(du - mu, mu - dl)
        open_ = np.ones(m, bool); open_[[r for r, _ in W]] = False
        so = np.where(open_, s, np.inf)
        r1 = int(np.argmin(so)); s1 = so[r1]
        for name, eps in PREC.items():
            E = eps * nu; st = stat[name]
            if final:
                st["final"] += 1
                inb = int((so < -tol * sc + E).sum())       # rows that do not clear their threshold by more than E
                st["final_band"] += inb > 0; st["final_rows"] += inb
            else:
                st["scans"] += 1
                cand = int((so < s1 + 2 * E).sum())
                und = cand > 1 or not (s1 < -tol * sc[r1] - E)
                st["undecided"] += und; st["cand"] += cand
        return u
    for e in md.get_trace():
        r = abs(int(e)) - 1
        if e > 0:
            u = scan(W, False)
            mu = M[r] @ u
            W.append((r, 1 if du[r] - mu < mu - dl[r] else -1))
        else:
            W = [(a, b) for a, b in W if a != r]
    scan(W, True)
for name, st in stat.items():
    print(f"{name:44s}: {st['scans']} picking scans, undecided {st['undecided'] / st['scans']:.4f}, rows within the band per scan {st['cand'] / st['scans']:.2f} | "
          f"{st['final']} final scans, with rows inside the band {st['final_band'] / st['final']:.4f} ({st['final_rows'] / st['final']:.2f} rows each)")
