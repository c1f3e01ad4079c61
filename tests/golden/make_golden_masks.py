"""tests/golden/make_golden_masks.py -- golden_update_masks.npz: every daqp_update_ldp mask the reference accepts on this path,
written by the REFERENCE library (strict-IEEE build, oracle/_ref: build container only).

    setup_daqp -> daqp_solve -> { daqp_update_ldp(mask, changed arrays) -> daqp_solve } x STEPS

for every mask of MASKS (utils.c:58-221 runs each bit's step on its own; the reference's own binding builds these masks field by
field, interfaces/daqp-python/daqp.pyx:513-571), on four shapes -- config C1's, config C3's (simple bounds: the column scalings of
utils.c:447-452,491-496), one with equality and soft rows, one beyond 64 variables (the generic kernels) -- TRIALS problems each,
so that a batch of TRIALS problems can replay one sequence.  Stored per (shape, mask, trial): the base problem, per step the arrays
handed over and the reference's update flag, x, lam, fval, iter, exit flag and working set after the solve.

The fixture is data (inputs and the reference's outputs); tests/test_gpu_update_masks.py replays it through Model, BatchModel and the
compiled C caller, tests/test_cpu.py::test_oracle_golden_update_masks through the oracle.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
R, M, V, D, S = O.UPDATE_Rinv, O.UPDATE_M, O.UPDATE_v, O.UPDATE_d, O.UPDATE_sense
MASKS = [S, M, M | D, R, D | S, V | D | S, R | M, R | V, M | V, R | M | V | D, M | S, R | S, M | V | D | S, R | D, V | S,
         R | M | V | D | S, V, D, V | D, R | M | D | S]
SHAPES = {  # name: (n, m, ms, n_active, masks, steps)
    "c1": (20, 40, 0, 8, MASKS, 3),
    "c3": (12, 48, 12, 6, MASKS, 3),
    "mix": (10, 30, 4, 5, MASKS, 3),
    "wide": (66, 100, 4, 20, [S, M, M | D, R, R | M | V | D], 2),
}
TRIALS = 3


def base_problem(name, trial):
    n, m, ms, na, _, _ = SHAPES[name]
    q = O.generate_qp(n, m, ms, na, rng=[2025, sum(map(ord, name)), trial])
    sense = np.zeros(m, np.int32)
    if name == "mix":          # an equality row (marked), soft rows among the general rows and, on one trial, a soft simple bound
        q["blower"][ms + 1] = q["bupper"][ms + 1]
        sense[ms + 1] = 5
        sense[ms + 3] = 8
        sense[m - 1] = 8
        if trial == 2:
            sense[1] = 8
    q["sense"] = sense
    return q


def step_arrays(name, trial, mask, step, q, last_lam):
    """the arrays the caller hands over with `mask` at this step (deterministic in (shape, trial, mask, step))"""
    n, m, ms = SHAPES[name][:3]
    r = np.random.default_rng([77, sum(map(ord, name)), trial, mask, step])
    kw = {}
    if mask & R:
        G = 0.1 * r.standard_normal((n, n))
        kw["H"] = q["H"] + G @ G.T
    if mask & M:
        kw["A"] = q["A"] + 0.05 * r.standard_normal(q["A"].shape)
    if mask & V:
        kw["f"] = q["f"] + 0.3 * r.standard_normal(n)
    if mask & D:
        w = 0.05 * r.random(m)
        kw["bupper"], kw["blower"] = q["bupper"] + w, q["blower"] - 0.5 * w
        if name == "mix":
            kw["blower"][ms + 1] = kw["bupper"][ms + 1]
            if step == 1:      # an UNMARKED equality appears: the bound check marks and activates it (utils.c:558-562)
                kw["blower"][ms + 5] = kw["bupper"][ms + 5]
            if step == 0 and trial == 1 and mask != R | M | V | D | S:
                # crossed bounds: the update ends with -1 at the bound check (utils.c:95-96) and the following daqp_solve runs on
                # what the workspace held before (an unmarked equality in front of the crossed pair has been marked by then);
                # the next step repairs the bounds.  (Not with every bit set: there the batch path is a re-setup, whose failed
                # bound check leaves no workspace -- INTEGRATION.md.)
                kw["blower"][ms + 2] = kw["bupper"][ms + 2]
                kw["bupper"][ms + 7] = kw["blower"][ms + 7] - 1.0
    if mask & S:
        s2 = q["sense"].copy()
        act = np.nonzero(last_lam)[0]
        if step == 0 and act.size:   # warm start from (part of) the last active set, with its side
            for j in act[: 1 + trial]:
                s2[j] |= 1 | (2 if last_lam[j] < 0 else 0)
        if step == 1:                # an arbitrary row marked active, upper side
            s2[int(r.integers(0, m))] |= 1
        kw["sense"] = s2             # step 2: the base sense again
    return kw


def main():
    ref = O.Reference(strict=True)
    out = {}
    count = 0
    for name, (n, m, ms, na, masks, steps) in SHAPES.items():
        for trial in range(TRIALS):
            q = base_problem(name, trial)
            for k in ("H", "f", "A", "bupper", "blower", "sense"):
                out[f"{name}/{trial}/{k}"] = q[k]
            for mask in masks:
                rm = ref.model(n, m, ms)
                assert rm.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"]) == 1
                x, lam, fval, flag, it = rm.solve()
                pre = f"{name}/{trial}/{mask}"
                out[f"{pre}/x0"], out[f"{pre}/lam0"] = x, lam
                out[f"{pre}/res0"] = np.array([fval, flag, it])
                out[f"{pre}/ws0"] = rm.working_set()
                for step in range(steps):
                    kw = step_arrays(name, trial, mask, step, q, lam)
                    uflag = rm.update(mask, **kw)
                    x, lam, fval, flag, it = rm.solve()
                    sp = f"{pre}/{step}"
                    for k, v in kw.items():
                        out[f"{sp}/{k}"] = v
                    out[f"{sp}/x"], out[f"{sp}/lam"] = x, lam
                    out[f"{sp}/res"] = np.array([fval, flag, it, uflag])
                    out[f"{sp}/ws"] = rm.working_set()
                    count += 1
                rm.close()
    out["masks"] = np.array(MASKS, np.int32)
    np.savez_compressed(os.path.join(HERE, "golden_update_masks.npz"), **out)
    print("wrote", count, "update + solve steps;", os.path.getsize(os.path.join(HERE, "golden_update_masks.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
