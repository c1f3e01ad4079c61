"""One-off large parity campaign (exact mode: bit-identical to the oracle; default mode: north_star bar).
usage: python tools/parity_campaign.py [nC2] [nC3] [nshapes] [nbig]
DAQP_CAMPAIGN_SALT=<k> (default 0) moves every draw (the configs' start index, the shape generator and the per-shape seeds) to fresh ones.
After the random shapes: the boundary shapes between the kernel families (n = 62 .. 66 around the 64-row working-set limit of the
register kernels, m around the 64-row block edges)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daqp_amd
from oracle import oracle as O

nC2 = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nC3 = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
nsh = int(sys.argv[3]) if len(sys.argv) > 3 else 150
nbig = int(sys.argv[4]) if len(sys.argv) > 4 else 40
os.environ["DAQP_AMD_NO_RECHECK"] = "1"      # every problem here has an optimum: the default-mode kernels' own verdicts, unassisted
ora = O.Oracle()
SALT = int(os.environ.get("DAQP_CAMPAIGN_SALT", "0"))


def bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


def run(tag, q, ms, exact):
    os.environ["DAQP_AMD_EXACT"] = "1" if exact else "0"
    g = daqp_amd.solve_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q.get("sense"), ms=ms)
    r = ora.quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q.get("sense"), ms=ms)
    same_flag = (g["exitflag"] == r[3]); same_it = (g["iter"] == r[4]); same_as = (np.sign(g["lam"]) == np.sign(r[1])).all(axis=1)
    ok = g["exitflag"] > 0
    dx = np.abs(g["x"][ok] - r[0][ok]).max(initial=0)
    bit = bits(g["x"][ok], r[0][ok]) and bits(g["lam"][ok], r[1][ok]) if exact else None
    print(f"{tag:28s} exact={int(exact)} N={len(same_flag):6d} flags {same_flag.mean():.6f} iters {same_it.mean():.6f} active sets {same_as.mean():.6f} "
          f"max|dx| {dx:.2e} bitwise {bit}", flush=True)
    return same_flag.all() and same_it.all() and same_as.all() and dx < 1e-9 and (bit is not False)


allok = True
for cfg, N in (("C2", nC2), ("C3", nC3)):
    n, m, ms, na, seed, _ = O.CONFIGS[cfg]
    q = O.generate_batch(N, n, m, ms, na, seed, start=500000 + 1000000 * SALT)
    for exact in (True, False):
        allok &= run(cfg, q, ms, exact)
rng = np.random.default_rng(2024 + SALT)
for s in range(nsh):
    n = int(rng.integers(2, 64)); m = int(rng.integers(n + 1, min(192, 4 * n + 8))); ms = int(rng.integers(0, min(n, m // 2) + 1))
    na = int(rng.integers(1, max(2, min(n, m - ms))))
    q = O.generate_batch(48, n, m, ms, na, 7000 + s + 100000 * SALT)
    for exact in (True, False):
        ok = run(f"shape n={n} m={m} ms={ms} na={na}", q, ms, exact)
        allok &= ok
# shapes beyond the full-register kernels with at most 64 working-set rows (round 6: the image-only kernels (4,32) (8,16) (6,25) (5,32), the (4,8) registers,
# n = 64 with 65 rows at a full vertex): n <= 64 with 129 ... 520 rows; some with nearly every row active
for s in range(int(os.environ.get("DAQP_CAMPAIGN_WIDE", "120"))):
    n = int(rng.integers(4, 65)); m = int(rng.integers(max(n + 1, 129), 521)); ms = int(rng.integers(0, min(n, m // 2) + 1))
    na = int(rng.integers(1, max(2, min(n, m - ms)))) if s % 5 else min(n - 1, m - ms)
    q = O.generate_batch(24, n, m, ms, na, 13000 + s + 100000 * SALT)
    for exact in (True, False):
        allok &= run(f"wide n={n} m={m} ms={ms} na={na}", q, ms, exact)
# shapes of the workgroup kernel / the generic setup with its own M launch (n > 64; working sets beyond 64 rows)
for s in range(nbig):
    n = int(rng.integers(65, 209)); m = int(rng.integers(n + 1, min(640, 3 * n + 8))); ms = int(rng.integers(0, min(n, m // 3) + 1))
    na = int(rng.integers(2, max(3, min(n - 1, (m - ms) // 2, 150))))
    q = O.generate_batch(6, n, m, ms, na, 9000 + s + 100000 * SALT)
    for exact in (True, False):
        allok &= run(f"shape n={n} m={m} ms={ms} na={na}", q, ms, exact)
# boundary shapes: the last register shape (n = 63: working sets of up to 64 rows), the first shapes beyond it (n = 64, 65: the workgroup kernel),
# row counts on either side of the 64-row blocks; some with nearly every row active (working sets that reach n + 1)
for n in (62, 63, 64, 65, 66):
    for m in (n + 1, 128, 129, 192, 193):
        for na in (4, n // 2, n - 1):
            ms = 0 if na != n // 2 else min(8, m - n)
            q = O.generate_batch(12, n, m, ms, min(na, m - ms), 11000 + 10 * n + m + 100000 * SALT)
            for exact in (True, False):
                allok &= run(f"boundary n={n} m={m} ms={ms} na={na}", q, ms, exact)
print("ALL OK" if allok else "MISMATCHES FOUND")
