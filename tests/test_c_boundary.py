"""A compiled C caller of the drop-in boundary: tests/c/boundary_caller.c is built with gcc against include/daqp_amd.h and
linked with -ldaqp_amd -- the reference's canonical usage (docs/docs/c.md, .github/benchmarks/maros_meszaros_runner.c:132-180).
Compile-time: _Static_asserts on sizeof / offsetof of the four structs.  Without a GPU the program must fail LOUDLY
(exit flag -8, "no HIP device"); on the GPU box its output is compared with the reference's own outputs (tests/golden)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "boundary_caller.c")


def build_caller(tmp_path):
    import daqp_amd
    daqp_amd.lib()   # (builds libdaqp_amd.so if it is missing)
    exe = os.path.join(str(tmp_path), "boundary_caller")
    libdir = os.path.join(ROOT, "daqp_amd", "lib")
    rocm = "/opt/rocm/lib"
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-L" + libdir, "-ldaqp_amd",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link," + rocm, "-Wl,-rpath," + rocm, "-o", exe]
    subprocess.check_call(cmd)
    return exe


def write_problem(path, H, f, A, bu, bl, sense, fs):
    n, m = f.size, bu.size
    A = np.ascontiguousarray(A, np.float64).reshape(-1, n)
    ms = m - A.shape[0]
    fs = np.ascontiguousarray(fs, np.float64).reshape(-1, n)
    with open(path, "wb") as fp:
        fp.write(struct.pack("4i", n, m, ms, fs.shape[0]))
        for a in (H, f, A, bu, bl):
            fp.write(np.ascontiguousarray(a, np.float64).tobytes())
        fp.write(np.ascontiguousarray(sense if sense is not None else np.zeros(m), np.int32).tobytes())
        fp.write(fs.tobytes())


def parse(out):
    res = {"solves": []}
    lines = out.strip().splitlines()
    vec = lambda ln: np.array([float.fromhex(t) for t in ln.split()[1:]])
    i = 0
    while i < len(lines):
        t = lines[i].split()
        if t[0] == "quadprog":
            res["flag"], res["iter"], res["nodes"] = int(t[2]), int(t[4]), int(t[6])
        elif t[0] == "error":
            res["error"] = lines[i][6:]
        elif t[0] == "fval":
            res["fval"] = float.fromhex(t[1])
        elif t[0] == "x" and "x" not in res:
            res["x"] = vec(lines[i])
        elif t[0] == "lam":
            res["lam"] = vec(lines[i])
        elif t[0] == "setup":
            res["setup"] = int(t[1])
        elif t[0] == "update":
            res.setdefault("updates", []).append(int(t[1]))
        elif t[0] == "solve":
            res["solves"].append(dict(flag=int(t[3]), iter=int(t[5]), n_active=int(t[7]), x=vec(lines[i + 1])))
            i += 1
        i += 1
    return res


def warm_fixture():
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_warm.npz"), allow_pickle=False)
    return g


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_caller_compiles_links_and_fails_loudly_without_a_gpu(tmp_path):
    """-m "not gpu": the header compiles as C11 with the struct-ABI static asserts, the program links against
    libdaqp_amd.so, and with no HIP device every solve reports DAQP_EXIT_UNSUPPORTED with a reason (no CPU fallback)"""
    import daqp_amd
    exe = build_caller(tmp_path)
    g = warm_fixture()
    prob = os.path.join(str(tmp_path), "p.bin")
    write_problem(prob, g["H"], g["f0"], g["A"], g["bupper"], g["blower"], None, g["fs"][1:])
    r = subprocess.run([exe, prob], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    res = parse(r.stdout)
    if daqp_amd.lib().daqp_amd_device_count() < 1:
        assert res["flag"] == -8 and "no HIP device" in res.get("error", ""), r.stdout
        assert res["setup"] == -8


@pytest.mark.gpu
@pytest.mark.parametrize("exact", ["1", "0"])
def test_c_caller_reproduces_the_reference_outputs(tmp_path, gpu_lib, exact):
    """the C program's daqp_quadprog and setup/solve/update sequence against the reference's own outputs (golden_warm.npz,
    golden_quadprog.npz): bit for bit in exact mode, flag / iterations / 1e-9 in the default arithmetic"""
    exe = build_caller(tmp_path)
    env = dict(os.environ, DAQP_AMD_EXACT=exact)
    g = warm_fixture()
    prob = os.path.join(str(tmp_path), "p.bin")
    write_problem(prob, g["H"], g["f0"], g["A"], g["bupper"], g["blower"], None, g["fs"][1:])
    r = subprocess.run([exe, prob], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    res = parse(r.stdout)
    assert res["flag"] == 1 and res["setup"] == 1 and res["nodes"] == 1, r.stdout
    assert res["updates"] == [0] * (g["fs"].shape[0] - 1)
    assert len(res["solves"]) == g["fs"].shape[0]
    for t, s in enumerate(res["solves"]):
        assert s["flag"] == int(g["exitflag"][t]) and s["iter"] == int(g["iter"][t]), (t, s)
        if exact == "1":
            assert np.array_equal(s["x"].view(np.uint64), g["x"][t].view(np.uint64)), t
        else:
            assert np.abs(s["x"] - g["x"][t]).max() < 1e-9, t
    assert res["solves"][0]["n_active"] > 0
    # daqp_quadprog leg on a few golden cases (hand example, config samples, a degenerate one with sense bits)
    q = np.load(os.path.join(ROOT, "tests", "golden", "golden_quadprog.npz"), allow_pickle=False)
    for nm in ("edge_py_demo", "C1_00", "C2_01", "C3_05", "nasty_03"):
        get = lambda f: q[f"{nm}/{f}"]
        sense = get("sense") if f"{nm}/sense" in q.files else None
        write_problem(prob, get("H"), get("f"), get("A"), get("bupper"), get("blower"), sense, np.zeros((0, get("f").size)))
        r = subprocess.run([exe, prob], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr
        res = parse(r.stdout)
        assert res["flag"] == int(get("exitflag")) and res["iter"] == int(get("iter")), (nm, r.stdout[:300])
        if res["flag"] > 0:
            if exact == "1":
                assert np.array_equal(res["x"].view(np.uint64), get("x").view(np.uint64)), nm
                assert np.array_equal(res["lam"].view(np.uint64), get("lam").view(np.uint64)) and res["fval"] == float(get("fval")), nm
            else:
                assert np.abs(res["x"] - get("x")).max() < 1e-9 and np.array_equal(np.sign(res["lam"]), np.sign(get("lam"))), nm
