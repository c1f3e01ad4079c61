"""Debug aid: one (shape, trial, mask) sequence of tests/golden/golden_update_masks.npz through the single-problem workspace with the
event trace switched on, next to the oracle's trace.  usage: debug_masks.py shape trial mask [exact]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
shape, trial, mask = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
os.environ["DAQP_AMD_EXACT"] = sys.argv[4] if len(sys.argv) > 4 else "0"
os.environ["DAQP_AMD_NO_RECHECK"] = "1"
os.environ["DAQP_AMD_NO_POOL"] = "1"
import daqp_amd  # noqa: E402
import mask_replay as MR  # noqa: E402
from oracle import oracle as O  # noqa: E402

L = daqp_amd.lib()
n, m, ms = MR.SHAPES[shape]
q = MR.base(shape, trial)
d = daqp_amd.Model()
assert d.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])[0] == 1
h = C.c_void_p(C.c_void_p.from_buffer(d._ws, 272).value)
CAP = 4096
L.daqp_batch_enable_trace(h, CAP)
om = O.Oracle().model(n, m, ms, ns=MR.NS[shape])
om.enable_trace()
om.setup(q["H"], q["f"], q["A"], q["bupper"], q["blower"], q["sense"])
steps = dict(MR.sequences(shape))[mask]
tlen = 0
for s in range(-1, steps):
    kw, exp = MR.step(shape, trial, mask, s)
    if s >= 0:
        print("update", d.update_mask(mask, **kw), om.update(mask, **kw), "ref", exp["uflag"])
        sense = np.zeros(m, np.int32)
        ws_sense = C.cast(C.c_void_p.from_buffer(d._ws, 64).value, C.POINTER(C.c_int))
        print("  sense gpu", [ws_sense[i] for i in range(m)])
        print("  sense ora", list(om.state()[1]))
    x, fval, ef, info = d.solve()
    r = om.solve()
    t = np.zeros(CAP, np.int32)
    L.daqp_batch_read_trace(h, t.ctypes.data_as(C.POINTER(C.c_int)))
    tg = t[: t[-1]]
    to = om.get_trace(marks=True)
    print(f"step {s}: gpu flag {ef} iter {info['iterations']} | oracle {r[3]} {r[4]} | ref {exp['flag']} {exp['iter']} | dx {np.abs(x - r[0]).max():.2e}")
    print("   gpu   ", list(tg[tlen_g:] if (tlen_g := globals().get('tlen_g', 0)) or True else []))
    print("   oracle", list(to[tlen:]))
    tlen = len(to)
    globals()['tlen_g'] = len(tg)
    M1, R1, v1, du1, dl1, sc1 = om.ldp()
    M2 = np.zeros_like(M1); R2 = np.zeros_like(R1); v2 = np.zeros_like(v1); du2 = np.zeros_like(du1); dl2 = np.zeros_like(dl1); sc2 = np.zeros_like(sc1)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    L.daqp_batch_read_ldp(h, 0, dp(M2), dp(R2), dp(v2), dp(du2), dp(dl2), dp(sc2))
    print("   ldp diff M %.1e R %.1e v %.1e du %.1e dl %.1e sc %.1e" % tuple(np.abs(a - b).max() if a.size else 0 for a, b in ((M1, M2), (R1, R2), (v1, v2), (du1, du2), (dl1, dl2), (sc1, sc2))))
