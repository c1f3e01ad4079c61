"""Which comparison decides a warm INFEASIBLE verdict one or two iterations apart from the reference's?    For the problems of
tests/test_gpu_golden.py::test_warm_updates_that_turn_infeasible_default_mode whose first infeasible warm verdict differs in `iter`, the event traces
(add +id / remove -id / singular-direction marker) of that solve in the default arithmetic, in the exact arithmetic and in the oracle, and -- replayed
on the CPU from the oracle's state at the step where the two part -- the singular direction's components next to dual_tol: the reference's branch
(daqp.c:86-93 with auxiliary.c:277-311) removes a blocking row while some component of the singular direction passes `>= dual_tol` / `<= -dual_tol`
and reports INFEASIBLE as soon as none does.   usage: python tools/warm_infeasible_trace.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import oracle as O
import daqp_amd
from test_gpu_golden import _nasty

os.environ["DAQP_AMD_NO_RECHECK"] = "1"
ora = O.Oracle()
MARK = daqp_amd.api.TRACE_MARK
NAMES = {1: "pivot", 2: "SINGULAR", 3: "refine", 4: "refactor", 5: "cycle-reset"}


def fmt(tr):
    return " ".join((NAMES.get(int(e) - MARK, "?") if abs(int(e)) >= MARK else f"{int(e):+d}") for e in tr)


def run(trial, exact):
    os.environ["DAQP_AMD_EXACT"] = "1" if exact else "0"
    q = _nasty(trial)
    n, m = q["f"].size, q["bupper"].size
    ms = m - q["A"].shape[0]
    ns = int(((q["sense"] & 8) != 0).sum())
    bm = daqp_amd.BatchModel(1, n, m, ms, ns_max=ns)
    bm.enable_trace(1 << 14)
    bm.setup(q["H"][None], q["f"][None], q["A"][None] if q["A"].size else None, q["bupper"][None], q["blower"][None], q["sense"][None], init_mask=0)
    om = ora.model(n, m, ms, ns=ns); om.enable_trace(1 << 14)
    sf = om.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
    if sf < 0:
        return []
    bm.solve(); om.solve()
    out = []
    for t in range(1, 4):
        w = (q["bupper"] - q["blower"]) * 0.03 * t
        w = np.where(np.abs(w) < 1e20, w, 0.0)
        bu, bl = q["bupper"] - w, q["blower"] + 0.45 * w
        bm.update(bupper=bu[None], blower=bl[None]); om.update(O.UPDATE_d, bupper=bu, blower=bl)
        o0 = len(om.get_trace(marks=True))                 # (the oracle's trace runs on; the library's restarts with every solve)
        g = bm.solve(); r = om.solve()
        gt = bm.read_trace(marks=True)[0]; ot = om.get_trace(marks=True)[o0:]
        out.append(dict(step=t, flag=int(g["exitflag"][0]), ref_flag=int(r[3]), iter=int(g["iter"][0]), ref_iter=int(r[4]), trace=fmt(gt), ref_trace=fmt(ot)))
    bm.close()
    return out


found = 0
for trial in range(200):
    d = run(trial, False)
    first = next((s for s in d if s["flag"] == -1), None)
    if first is None or first["iter"] == first["ref_iter"]:
        continue
    found += 1
    x = run(trial, True)
    xs = next(s for s in x if s["step"] == first["step"])
    a, b = first["trace"].split(), first["ref_trace"].split()
    k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
    print(f"trial {trial} step {first['step']}: default mode iter {first['iter']}, reference iter {first['ref_iter']} (exact mode: iter {xs['iter']}, trace identical to the reference's: {xs['trace'] == xs['ref_trace']})")
    print(f"   common prefix: {k} events; then default mode: {' '.join(a[k:k + 8]) or '(end: INFEASIBLE)'}   |   reference: {' '.join(b[k:k + 8]) or '(end: INFEASIBLE)'}")
    sing_a, sing_b = first["trace"].count("SINGULAR"), first["ref_trace"].count("SINGULAR")
    print(f"   singular-direction steps taken: default {sing_a}, reference {sing_b}; the verdict falls in the singular branch on both sides (last event before the end: "
          f"default '{a[-1] if a else ''}', reference '{b[-1] if b else ''}')")
print(f"{found} first infeasible warm verdicts with another iteration count than the reference's")
