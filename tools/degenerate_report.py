"""tools/degenerate_report.py -- which decision flips on the degenerate problems that the default arithmetic solves along another
path than the reference (tests/test_gpu_golden.py::test_degenerate_cases_fast_mode writes gpurun_out/degenerate_fast_mode.json
on the GPU box).  For every differing trial the oracle is stopped (iter_limit) right after the iteration in which the two paths
part and its raw iterate is read: the quantity next to the threshold that decided.  Runs on the CPU (oracle only).

    python tools/degenerate_report.py gpurun_out/degenerate_fast_mode.json profiles/r03_degenerate_fast_mode.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402


def nasty(trial):   # tests/test_gpu_golden.py::_nasty
    rng = np.random.default_rng([99, trial])
    eps = 10.0 ** rng.uniform(-13, -2)
    n = int(rng.integers(4, 16)); m = int(rng.integers(n + 4, 4 * n)); ms = int(rng.integers(0, min(n, m // 3) + 1))
    na = int(rng.integers(1, min(n, m - ms)))
    return eps, O.generate_nasty(n, m, ms, na, eps, rng, n_dup=int(rng.integers(0, 5)), n_eq=int(rng.integers(0, 3)),
                                 n_soft=int(rng.integers(0, 3)), dep_eq=bool(rng.integers(0, 2)))


def main(src, dst):
    d = json.load(open(src))
    ora = O.Oracle()
    ora.lib.ora_get_iterate.argtypes = [C.c_void_p, C.c_int, O.c_double_p, O.c_double_p]
    for rec in d["differing"]:
        eps, q = nasty(rec["trial"])
        n, m = q["f"].size, q["bupper"].size
        ms = m - q["A"].reshape(-1, n).shape[0]
        ns = int(((q["sense"] & O.SOFT) != 0).sum())
        lim = min(rec["iter"], rec["ref_iter"]) + 1       # the oracle right after the iteration in which the GPU path ended / parted

        def run(limit):
            om = ora.model(n, m, ms, ns, settings=O.default_settings(iter_limit=limit))
            om.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"], init_mask=O.UPDATE_unconstrained)
            om.enable_trace()
            om.solve()
            WS, sense, D, sing = om.state()
            lam, ls = np.zeros(n + ns + 2), np.zeros(n + ns + 2)
            ora.lib.ora_get_iterate(om.h, n + ns + 2, lam.ctypes.data_as(O.c_double_p), ls.ctypes.data_as(O.c_double_p))
            return WS, D, sing, ls, om.get_trace(marks=True)

        WS0, D0, sing0, _, _ = run(lim - 1)               # entering that iteration
        WS1, D1, sing1, ls, tr = run(lim)                 # leaving it (lam_star still holds that iteration's direction)
        rec["eps_of_the_duplicates"] = eps
        rec["oracle_last_events"] = [int(e) for e in tr[-4:]]
        rec["oracle_pivots_entering_the_iteration"] = [float(v) for v in D0]
        if len(tr) >= 2 and tr[-2] == O.TRACE_MARK + 2 and tr[-1] < 0 and sing0 >= 0:
            # a singular step (auxiliary.c:357-376) followed by the removal of a "blocking" row (auxiliary.c:277-311)
            rid = -int(tr[-1]) - 1
            pos = int(np.where(WS0 == rid)[0][0])
            comp = float(ls[pos])
            dpos = float(D0[D0 > 0].min())
            rec["decision"] = ("daqp_remove_blocking after daqp_compute_singular_direction (auxiliary.c:284-287, 357-376): the component of the "
                               f"singular direction of constraint {rid} (position {pos} of the working set) is {comp:.3e} in the reference's arithmetic "
                               f"-- 0 in exact arithmetic: it is the rounding noise of a back-substitution through the pivot D = {dpos:.2e} of a "
                               f"duplicated row (eps = {eps:.1e}) -- and is compared with dual_tol = 1e-12: the reference sees a blocking row, drops "
                               "it, finds the factor singular again and reports infeasibility one iteration later; with fused multiply-adds the noise "
                               "falls within dual_tol (or has the other sign), no row blocks, and the same infeasibility is reported in this "
                               "iteration.  Same exit flag (-1); iter differs by one; no x is returned either way.")
            rec["singular_direction_component"] = comp
    d["note"] = ("default arithmetic (fused multiply-adds, M = A R^-1 on the matrix cores) against the oracle on the 400 near-degenerate "
                 "problems of tests/test_gpu_golden.py; `differing` lists every trial whose exit flag, iteration count or active set is "
                 "not identical, with the decision that fell the other way")
    json.dump(d, open(dst, "w"), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
