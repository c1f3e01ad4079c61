"""the compiled C caller of tests/c/mask_caller.c run over and over (a rare crash of the single-problem path shows up as a non-zero exit code
with the call stack its signal handler prints): tools/stress_caller.py [runs per sequence] [DAQP_AMD_EXACT]"""
import os, subprocess, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_update_masks as T
import mask_replay as MR
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
exact = sys.argv[2] if len(sys.argv) > 2 else "1"
only = sys.argv[3] if len(sys.argv) > 3 else ""       # e.g. "c1" : that shape's sequences only
extra = dict(kv.split("=", 1) for kv in sys.argv[4:])  # further environment, KEY=VALUE ...
tmp = pathlib.Path(tempfile.mkdtemp())
exe = T._build_mask_caller(tmp)
env = dict(os.environ, DAQP_AMD_EXACT=exact, **extra)
bad = 0
for shape, trial, mask in (("c1", 0, T.M), ("c1", 1, T.R), ("c3", 2, T.M | T.D), ("c3", 0, T.S), ("mix", 1, T.R | T.D), ("mix", 1, T.M | T.V | T.D | T.S),
                           ("mix", 2, T.R | T.S), ("wide", 0, T.M), ("c3", 1, T.R | T.M | T.D | T.S)):
    if only and shape != only: continue
    steps = dict(MR.sequences(shape))[mask]
    seq = str(tmp / "seq.bin")
    T._write_sequence(seq, shape, trial, mask, steps)
    ref = None
    for i in range(runs):
        try:
            r = subprocess.run([exe, seq], capture_output=True, text=True, timeout=60, env=env)
        except subprocess.TimeoutExpired:
            bad += 1
            print("HANG", shape, trial, mask, "run", i, flush=True)
            continue
        if r.returncode != 0:
            bad += 1
            print("FAIL", shape, trial, mask, "run", i, "rc", r.returncode, "\n", r.stderr[-1500:], "\nstdout: lines", len(r.stdout.splitlines()), "last:", r.stdout.splitlines()[-1:] , flush=True)
        elif ref is None:
            ref = r.stdout
        elif r.stdout != ref:
            bad += 1
            print("DIFFERENT OUTPUT", shape, trial, mask, "run", i, flush=True)
    print("done", shape, trial, mask, flush=True)
print("failures:", bad)
