"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (see oracle/daqp_oracle.c).

ctypes drivers for
  * ``Oracle``    -- this repository's C restatement (oracle/liboracle.so)
  * ``Reference`` -- the reference library itself, built by oracle/Makefile into
                     oracle/_ref/ from /root/reference (never copied here)
and the numpy restatement of the reference's random-QP generator
(interfaces/daqp-julia/test/utils.jl:3-53).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)

# update masks / exit flags (reference constants.h:42-61)
UPDATE_Rinv, UPDATE_M, UPDATE_v, UPDATE_d, UPDATE_sense = 1, 2, 4, 8, 16
UPDATE_unconstrained, UPDATE_eliminate = 64, 128
ACTIVE, LOWER, IMMUTABLE, SOFT = 1, 2, 4, 8
TRACE_MARK = 0x40000000


class Settings(C.Structure):
    """Byte-compatible with DAQPSettings (reference types.h:52-74)."""
    _fields_ = [("primal_tol", C.c_double), ("dual_tol", C.c_double), ("zero_tol", C.c_double),
                ("pivot_tol", C.c_double), ("progress_tol", C.c_double),
                ("cycle_tol", C.c_int), ("iter_limit", C.c_int),
                ("fval_bound", C.c_double), ("eps_prox", C.c_double), ("eta_prox", C.c_double),
                ("rho_soft", C.c_double), ("rel_subopt", C.c_double), ("abs_subopt", C.c_double),
                ("sing_tol", C.c_double), ("refactor_tol", C.c_double), ("time_limit", C.c_double)]


def default_settings(**kw):
    s = Settings(1e-6, 1e-12, 1e-11, 1e-6, 1e-14, 10, 10000, 1e30, -1e-6, -1.0, 1e-6, 0, 0,
                 3.7e-11, 1e-9, 0)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def build(force=False):
    """Compile liboracle*.so (and oracle/_ref when /root/reference is present)."""
    need = force or not os.path.exists(os.path.join(HERE, "liboracle.so")) \
        or os.path.getmtime(os.path.join(HERE, "liboracle.so")) < os.path.getmtime(os.path.join(HERE, "daqp_oracle.c")) \
        or (os.path.exists(os.path.join(HERE, "librefbatch.so"))
            and os.path.getmtime(os.path.join(HERE, "librefbatch.so")) < os.path.getmtime(os.path.join(HERE, "ref_batch.c")))
    ref_missing = os.path.isdir("/root/reference/src") and not os.path.exists(os.path.join(HERE, "_ref", "libdaqp_ref.so"))
    if need or ref_missing or not os.path.exists(os.path.join(HERE, "liboracle_fast.so")) \
            or not os.path.exists(os.path.join(HERE, "librefbatch.so")):
        subprocess.check_call(["make", "-s", "-C", HERE, "all", os.path.join(HERE, "liboracle_fast.so")])


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_int_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


class Oracle:
    """The C restatement.  ``fast=True`` loads the build with the reference's release flags
    (used only for cpu_baseline timing)."""

    def __init__(self, fast=False):
        build()
        self.lib = C.CDLL(os.path.join(HERE, "liboracle_fast.so" if fast else "liboracle.so"))
        L = self.lib
        L.ora_create.restype = C.c_void_p
        L.ora_create.argtypes = [C.c_int] * 4 + [C.POINTER(Settings)]
        L.ora_free.argtypes = [C.c_void_p]
        L.ora_setup.argtypes = [C.c_void_p, C.c_int] + [c_double_p] * 5 + [c_int_p]
        L.ora_update.argtypes = [C.c_void_p, C.c_int] + [c_double_p] * 5 + [c_int_p]
        L.ora_solve.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, c_int_p, c_double_p]
        L.ora_quadprog.argtypes = [C.c_int] * 3 + [c_double_p] * 5 + [c_int_p, C.POINTER(Settings),
                                                                       c_double_p, c_double_p, c_double_p, c_int_p]
        L.ora_quadprog_batch.argtypes = [C.c_int] * 4 + [c_double_p] * 5 + [c_int_p, C.POINTER(Settings),
                                                                             c_double_p, c_double_p, c_double_p, c_int_p, c_int_p]
        L.ora_quadprog_batch.restype = None
        L.ora_set_trace.argtypes = [C.c_void_p, c_int_p, C.c_int]
        L.ora_trace_len.argtypes = [C.c_void_p]
        L.ora_get_state.argtypes = [C.c_void_p, c_int_p, c_int_p, c_int_p, c_double_p]
        L.ora_get_ldp.argtypes = [C.c_void_p] + [c_double_p] * 6
        L.ora_get_prox.argtypes = [C.c_void_p, c_int_p, c_int_p]
        L.ora_set_primal_start.argtypes = [C.c_void_p, c_double_p]
        L.ora_set_primal_start.restype = None

    def quadprog(self, H, f, A, bupper, blower, sense=None, settings=None):
        H, f, A, bupper, blower, sense = _f64(H), _f64(f), _f64(A), _f64(bupper), _f64(blower), _i32(sense)
        n, m = f.size, bupper.size
        ms = m - (A.shape[0] if A.ndim == 2 else A.size // n)
        x, lam = np.zeros(n), np.zeros(m)
        fval, it = C.c_double(0), C.c_int(0)
        st = settings if settings is not None else default_settings()
        flag = self.lib.ora_quadprog(n, m, ms, _dp(H), _dp(f), _dp(A), _dp(bupper), _dp(blower), _ip(sense),
                                     C.byref(st), _dp(x), _dp(lam), C.byref(fval), C.byref(it))
        return x, lam, fval.value, flag, it.value

    def quadprog_batch(self, H, f, A, bupper, blower, sense=None, settings=None, ms=0):
        H, f, A, bupper, blower, sense = _f64(H), _f64(f), _f64(A), _f64(bupper), _f64(blower), _i32(sense)
        N, n = f.shape
        m = bupper.shape[1]
        x, lam, fval = np.zeros((N, n)), np.zeros((N, m)), np.zeros(N)
        flag, it = np.zeros(N, np.int32), np.zeros(N, np.int32)
        st = settings if settings is not None else default_settings()
        self.lib.ora_quadprog_batch(N, n, m, ms, _dp(H), _dp(f), _dp(A), _dp(bupper), _dp(blower), _ip(sense),
                                    C.byref(st), _dp(x), _dp(lam), _dp(fval), _ip(flag), _ip(it))
        return x, lam, fval, flag, it

    def model(self, n, m, ms, ns=0, settings=None):
        return OracleModel(self, n, m, ms, ns, settings)


class OracleModel:
    """setup_daqp / daqp_update_ldp / daqp_solve sequence on one oracle workspace."""

    def __init__(self, ora, n, m, ms, ns=0, settings=None):
        self.o, self.n, self.m, self.ms = ora, n, m, ms
        st = settings if settings is not None else default_settings()
        self.h = ora.lib.ora_create(n, m, ms, ns, C.byref(st))
        self.keep = {}
        self.trace = None

    def __del__(self):
        if getattr(self, "h", None):
            self.o.lib.ora_free(self.h)
            self.h = None

    def enable_trace(self, cap=100000):
        self.trace = np.zeros(cap, np.int32)
        self.o.lib.ora_set_trace(self.h, _ip(self.trace), cap)

    def get_trace(self, marks=False):
        """+(id+1) add / -(id+1) remove in order; marks=True keeps the branch markers (TRACE_MARK + 1..5: pivot, singular
        direction, refine, refactor repair, cycle-guard rebuild -- the GPU kernels' traces carry the same codes)"""
        t = self.trace[: self.o.lib.ora_trace_len(self.h)].copy()
        return t if marks else t[t < TRACE_MARK]

    def _hold(self, **kw):
        for k, v in kw.items():
            if v is not None:
                self.keep[k] = v
        return [self.keep.get(k) for k in ("H", "f", "A", "bu", "bl", "sense")]

    def setup(self, H, f, A, bupper, blower, sense=None, init_mask=0):
        a = self._hold(H=_f64(H), f=_f64(f), A=_f64(A), bu=_f64(bupper), bl=_f64(blower), sense=_i32(sense))
        self.keep["sense"] = a[5]
        return self.o.lib.ora_setup(self.h, init_mask, _dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]), _dp(a[4]), _ip(a[5]))

    def update(self, mask, H=None, f=None, A=None, bupper=None, blower=None, sense=None):
        a = self._hold(H=_f64(H), f=_f64(f), A=_f64(A), bu=_f64(bupper), bl=_f64(blower), sense=_i32(sense))
        return self.o.lib.ora_update(self.h, mask, _dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]), _dp(a[4]), _ip(a[5]))

    def solve(self):
        x, lam = np.zeros(self.n), np.zeros(self.m)
        fval, it, ss = C.c_double(0), C.c_int(0), C.c_double(0)
        flag = self.o.lib.ora_solve(self.h, _dp(x), _dp(lam), C.byref(fval), C.byref(it), C.byref(ss))
        return x, lam, fval.value, flag, it.value

    def state(self):
        na = C.c_int(0)
        WS, sense, D = np.zeros(self.n + self.m + 1, np.int32), np.zeros(self.m, np.int32), np.zeros(self.n + self.m + 1)
        sing = self.o.lib.ora_get_state(self.h, C.byref(na), _ip(WS), _ip(sense), _dp(D))
        return WS[: na.value].copy(), sense, D[: na.value].copy(), sing

    def set_primal_start(self, x):
        self.o.lib.ora_set_primal_start(self.h, _dp(_f64(x)))

    def prox(self):
        """(n_prox, outer iterations of the last solve, prox_mask)"""
        nh, mask = C.c_int(0), np.zeros(self.n, np.int32)
        npx = self.o.lib.ora_get_prox(self.h, C.byref(nh), _ip(mask))
        return npx, nh.value, mask

    def ldp(self):
        n, m, ms = self.n, self.m, self.ms
        M, R, v = np.zeros((m - ms, n)), np.zeros(n * (n + 1) // 2), np.zeros(n)
        du, dl, sc = np.zeros(m), np.zeros(m), np.zeros(m)
        self.o.lib.ora_get_ldp(self.h, _dp(M), _dp(R), _dp(v), _dp(du), _dp(dl), _dp(sc))
        return M, R, v, du, dl, sc


# ---------------------------------------------------------------------------
# the reference library (oracle/_ref), driven through its own C API
# ---------------------------------------------------------------------------
class _Problem(C.Structure):  # DAQPProblem, reference types.h:14-50 (80 bytes)
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("ms", C.c_int),
                ("H", c_double_p), ("f", c_double_p), ("A", c_double_p),
                ("bupper", c_double_p), ("blower", c_double_p), ("sense", c_int_p),
                ("break_points", c_int_p), ("nh", C.c_int), ("problem_type", C.c_int)]


class _Result(C.Structure):  # DAQPResult, reference api.h:15-27 (64 bytes)
    _fields_ = [("x", c_double_p), ("lam", c_double_p), ("fval", C.c_double), ("soft_slack", C.c_double),
                ("exitflag", C.c_int), ("iter", C.c_int), ("nodes", C.c_int),
                ("solve_time", C.c_double), ("setup_time", C.c_double)]


_WS_BYTES = 288          # sizeof(DAQPWorkspace), measured against the reference headers
_WS_SETTINGS_OFF = 224   # offsetof(DAQPWorkspace, settings)
_WS_NACTIVE_OFF, _WS_WS_OFF, _WS_SENSE_OFF = 184, 176, 64


def reference_available(strict=False):
    return os.path.exists(os.path.join(HERE, "_ref", "libdaqp_ref_strict.so" if strict else "libdaqp_ref.so"))


class Reference:
    """The reference C library.  strict=True: the -O2 strict-IEEE build (bit-exact pin target);
    strict=False: the reference's own release flags (cpu_baseline kind 'reference')."""

    def __init__(self, strict=False):
        build()
        path = os.path.join(HERE, "_ref", "libdaqp_ref_strict.so" if strict else "libdaqp_ref.so")
        self.lib = C.CDLL(path)
        L = self.lib
        L.daqp_quadprog.argtypes = [C.POINTER(_Result), C.POINTER(_Problem), C.POINTER(Settings)]
        L.daqp_quadprog.restype = None
        L.setup_daqp.argtypes = [C.POINTER(_Problem), C.c_void_p, c_double_p]
        L.daqp_update_ldp.argtypes = [C.c_int, C.c_void_p, C.POINTER(_Problem)]
        L.daqp_solve.argtypes = [C.POINTER(_Result), C.c_void_p]
        L.daqp_solve.restype = None
        L.free_daqp_workspace.argtypes = [C.c_void_p]
        L.free_daqp_ldp.argtypes = [C.c_void_p]
        L.daqp_default_settings.argtypes = [C.POINTER(Settings)]
        L.daqp_set_primal_start.argtypes = [C.c_void_p, c_double_p]
        L.daqp_set_primal_start.restype = None

    @staticmethod
    def _problem(H, f, A, bupper, blower, sense, ms):
        n, m = f.size, bupper.size
        return _Problem(n, m, ms, _dp(H), _dp(f), _dp(A), _dp(bupper), _dp(blower), _ip(sense), None, 0, 0)

    def quadprog(self, H, f, A, bupper, blower, sense=None, settings=None):
        H, f, A, bupper, blower, sense = _f64(H), _f64(f), _f64(A), _f64(bupper), _f64(blower), _i32(sense)
        n, m = f.size, bupper.size
        ms = m - (A.shape[0] if A.ndim == 2 else A.size // n)
        qp = self._problem(H, f, A, bupper, blower, sense, ms)
        x, lam = np.zeros(n), np.zeros(m)
        res = _Result(_dp(x), _dp(lam), 0, 0, 0, 0, 0, 0, 0)
        self.lib.daqp_quadprog(C.byref(res), C.byref(qp), C.byref(settings) if settings is not None else None)
        return x, lam, res.fval, res.exitflag, res.iter

    def quadprog_batch(self, H, f, A, bupper, blower, sense=None, settings=None, ms=0, timing=False):
        N = f.shape[0]
        out = [np.zeros_like(f), np.zeros_like(bupper), np.zeros(N), np.zeros(N, np.int32), np.zeros(N, np.int32)]
        ts = np.zeros((N, 2))
        for q in range(N):
            n, m = f.shape[1], bupper.shape[1]
            qp = self._problem(H[q], f[q], A[q], bupper[q], blower[q], None if sense is None else sense[q], ms)
            res = _Result(_dp(out[0][q]), _dp(out[1][q]), 0, 0, 0, 0, 0, 0, 0)
            self.lib.daqp_quadprog(C.byref(res), C.byref(qp), C.byref(settings) if settings is not None else None)
            out[2][q], out[3][q], out[4][q] = res.fval, res.exitflag, res.iter
            ts[q] = res.setup_time, res.solve_time
        return (*out, ts) if timing else tuple(out)

    def model(self, n, m, ms, settings=None):
        return ReferenceModel(self, n, m, ms, settings)


class ReferenceModel:
    """setup_daqp -> {daqp_update_ldp -> daqp_solve}* on one reference workspace (docs/docs/c.md:49-71)."""

    def __init__(self, ref, n, m, ms, settings=None):
        self.r, self.n, self.m, self.ms = ref, n, m, ms
        self.ws = C.create_string_buffer(_WS_BYTES + 64)
        self.settings = settings
        self.keep = {}
        self.qp = None
        self.live = False

    def _qp(self):
        k = self.keep
        self.qp = _Problem(self.n, self.m, self.ms, _dp(k["H"]), _dp(k["f"]), _dp(k["A"]), _dp(k["bu"]),
                           _dp(k["bl"]), _ip(k.get("sense")), None, 0, 0)
        return self.qp

    def setup(self, H, f, A, bupper, blower, sense=None):
        self.keep = dict(H=_f64(H), f=_f64(f), A=_f64(A), bu=_f64(bupper), bl=_f64(blower), sense=_i32(sense))
        if self.settings is not None:
            C.c_void_p.from_buffer(self.ws, _WS_SETTINGS_OFF).value = C.addressof(self.settings)
        t = C.c_double(0)
        flag = self.r.lib.setup_daqp(C.byref(self._qp()), self.ws, C.byref(t))
        self.live = flag >= 0
        return flag

    def update(self, mask, H=None, f=None, A=None, bupper=None, blower=None, sense=None):
        for k, v in dict(H=_f64(H), f=_f64(f), A=_f64(A), bu=_f64(bupper), bl=_f64(blower), sense=_i32(sense)).items():
            if v is not None:
                self.keep[k] = v
        return self.r.lib.daqp_update_ldp(mask, self.ws, C.byref(self._qp()))

    def solve(self):
        x, lam = np.zeros(self.n), np.zeros(self.m)
        res = _Result(_dp(x), _dp(lam), 0, 0, 0, 0, 0, 0, 0)
        self.r.lib.daqp_solve(C.byref(res), self.ws)
        return x, lam, res.fval, res.exitflag, res.iter

    def set_primal_start(self, x):
        x = _f64(x)
        self.r.lib.daqp_set_primal_start(C.byref(self.ws), _dp(x))

    def working_set(self):
        na = C.c_int.from_buffer(self.ws, _WS_NACTIVE_OFF).value
        p = C.cast(C.c_void_p.from_buffer(self.ws, _WS_WS_OFF).value, c_int_p)
        return np.array([p[i] for i in range(na)], np.int32)

    def close(self):
        if self.live:
            if self.settings is not None:  # borrowed settings must not be freed (api.c:66,75)
                C.c_void_p.from_buffer(self.ws, _WS_SETTINGS_OFF).value = None
            self.r.lib.free_daqp_workspace(self.ws)
            self.r.lib.free_daqp_ldp(self.ws)
            self.live = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# synthetic inputs: numpy restatement of generate_test_QP
# (reference interfaces/daqp-julia/test/utils.jl:3-53)
# ---------------------------------------------------------------------------
def generate_qp(n, m, ms, n_active, kappa=100.0, rng=None):
    rng = np.random.default_rng(rng)
    eig = np.empty(n)
    eig[0], eig[1] = 1.0, kappa
    eig[2:] = 1.0 + (kappa - 1.0) * rng.random(n - 2)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    T = np.sqrt(eig)[:, None] * Q.T
    Tinv = Q / np.sqrt(eig)[None, :]
    H = T.T @ T
    M = np.vstack([Tinv[:ms, :], rng.standard_normal((m - ms, n))])
    dupper, dlower = np.zeros(m), np.zeros(m)
    perm = rng.permutation(m)
    n_up = int(rng.integers(0, n_active + 1))
    ids_up, ids_lo, ids_in = perm[:n_up], perm[n_up:n_active], perm[n_active:]
    lam = rng.random(n_active)
    Ma = np.vstack([M[ids_up], -M[ids_lo]])
    da = -Ma @ (Ma.T @ lam)
    dupper[ids_up] = da[:n_up]
    dlower[ids_lo] = -da[n_up:]
    u = -Ma.T @ lam
    dupper[ids_lo] = dlower[ids_lo] + (0.01 + rng.random(ids_lo.size))
    dlower[ids_up] = dupper[ids_up] - (0.01 + rng.random(ids_up.size))
    dupper[ids_in] = M[ids_in] @ u + (0.01 + rng.random(ids_in.size))
    dlower[ids_in] = M[ids_in] @ u - (0.01 + rng.random(ids_in.size))
    v = rng.standard_normal(n)
    f = T.T @ v
    x = np.linalg.solve(T, u - v)
    A = M[ms:] @ T
    bupper, blower = dupper - M @ v, dlower - M @ v
    return dict(x=x, H=0.5 * (H + H.T), f=f, A=np.ascontiguousarray(A), bupper=bupper, blower=blower,
                sense=np.zeros(m, np.int32))


def generate_batch(N, n, m, ms, n_active, seed, kappa=100.0, start=0):
    """QP k of a config uses default_rng([seed, k])."""
    out = dict(H=np.empty((N, n, n)), f=np.empty((N, n)), A=np.empty((N, m - ms, n)),
               bupper=np.empty((N, m)), blower=np.empty((N, m)), xref=np.empty((N, n)))
    for k in range(N):
        q = generate_qp(n, m, ms, n_active, kappa, rng=[seed, start + k])
        out["H"][k], out["f"][k], out["A"][k] = q["H"], q["f"], q["A"]
        out["bupper"][k], out["blower"][k], out["xref"][k] = q["bupper"], q["blower"], q["x"]
    return out


def timed_cpu_batch(libpath, threads, H, f, A, bupper, blower, ms=0, passes=1):
    """daqp_quadprog of `libpath` over a batch on `threads` host threads (oracle/ref_batch.c), `passes` sweeps over the batch
    inside the clock (threads are created before it starts).  Returns (seconds of all passes, x, lam, fval, exitflag, iter)."""
    build()
    L = C.CDLL(os.path.join(HERE, "librefbatch.so"))
    L.ref_batch_run.restype = C.c_double
    L.ref_batch_run.argtypes = [C.c_char_p] + [C.c_int] * 6 + [c_double_p] * 8 + [c_int_p] * 2
    H, f, A, bupper, blower = _f64(H), _f64(f), _f64(A), _f64(bupper), _f64(blower)
    N, n = f.shape
    m = bupper.shape[1]
    x, lam, fval = np.zeros((N, n)), np.zeros((N, m)), np.zeros(N)
    flag, it = np.zeros(N, np.int32), np.zeros(N, np.int32)
    dt = L.ref_batch_run(libpath.encode(), threads, int(passes), N, n, m, ms, _dp(H), _dp(f), _dp(A), _dp(bupper), _dp(blower),
                         _dp(x), _dp(lam), _dp(fval), _ip(flag), _ip(it))
    if dt < 0:
        raise RuntimeError(f"could not load daqp_quadprog from {libpath}")
    return dt, x, lam, fval, flag, it


def timed_cpu_warm(libpath, threads, H, f, A, bupper, blower, fs, ms=0, passes=1):
    """config C5 on `threads` host threads, all in C (oracle/ref_batch.c::ref_warm_run): setup_daqp + cold daqp_solve per QP
    (untimed), then the T = fs.shape[0] warm steps daqp_update_ldp(UPDATE_v) + daqp_solve of every QP (timed).
    `passes` > 1 walks the same path back and forth inside the clock (ref_batch.c); outputs are those of the first, forward pass.
    Returns (seconds of the warm phase, x [T,N,n], lam [T,N,m], exitflag [T,N], iter [T,N])."""
    build()
    L = C.CDLL(os.path.join(HERE, "librefbatch.so"))
    L.ref_warm_run.restype = C.c_double
    L.ref_warm_run.argtypes = [C.c_char_p] + [C.c_int] * 7 + [c_double_p] * 8 + [c_int_p] * 2
    H, f, A, bupper, blower, fs = _f64(H), _f64(f), _f64(A), _f64(bupper), _f64(blower), _f64(fs)
    N, n = f.shape
    m, T = bupper.shape[1], fs.shape[0]
    x, lam = np.zeros((T, N, n)), np.zeros((T, N, m))
    flag, it = np.zeros((T, N), np.int32), np.zeros((T, N), np.int32)
    dt = L.ref_warm_run(libpath.encode(), threads, int(passes), N, n, m, ms, T, _dp(H), _dp(f), _dp(A), _dp(bupper), _dp(blower), _dp(fs),
                        _dp(x), _dp(lam), _ip(flag), _ip(it))
    if dt < 0:
        raise RuntimeError(f"could not load the workspace API from {libpath}")
    return dt, x, lam, flag, it


CONFIGS = {  # name -> (n, m, ms, n_active, seed, full N)
    "C1": (20, 40, 0, 8, 1234, 1),
    "C2": (50, 150, 0, 20, 42, 100_000),
    "C3": (12, 48, 12, 6, 43, 1_000_000),
    "C4": (200, 600, 0, 80, 44, 10_000),
}


def generate_nasty(n, m, ms, n_active, eps, rng, n_dup=3, n_eq=0, n_soft=0, dep_eq=False, kappa=100.0):
    """A generator QP made degenerate on purpose: near-duplicates (relative distance ``eps``) of rows
    that are active at the optimum, with slightly shifted bounds, so that LDL pivots get tiny
    (pivoting / singular / refinement / refactor branches of daqp.c and auxiliary.c); optional
    equality rows (sense 5), linearly dependent equalities, and soft rows (sense 8)."""
    rng = np.random.default_rng(rng)
    q = generate_qp(n, m, ms, n_active, kappa, rng=rng)
    A, bu, bl, x = q["A"].copy(), q["bupper"].copy(), q["blower"].copy(), q["x"]
    mA = m - ms
    Ax = A @ x
    act_up = [i for i in range(mA) if abs(Ax[i] - bu[ms + i]) < 1e-9]
    act_lo = [i for i in range(mA) if abs(Ax[i] - bl[ms + i]) < 1e-9]
    free = [i for i in range(mA) if i not in act_up and i not in act_lo]
    rng.shuffle(free)
    sense = np.zeros(m, np.int32)
    used = 0
    for src in (act_up + act_lo)[:n_dup]:
        if used >= len(free):
            break
        dst = free[used]; used += 1
        A[dst] = A[src] * (1.0 + eps * rng.standard_normal()) + eps * np.linalg.norm(A[src]) * rng.standard_normal(n) / np.sqrt(n)
        shift = eps * (rng.random() - 0.3)
        if src in act_up:
            bu[ms + dst] = A[dst] @ x - abs(shift); bl[ms + dst] = bu[ms + dst] - 1.0
        else:
            bl[ms + dst] = A[dst] @ x + abs(shift); bu[ms + dst] = bl[ms + dst] + 1.0
    for _ in range(n_eq):
        if used >= len(free):
            break
        dst = free[used]; used += 1
        val = A[dst] @ x + 0.05 * rng.standard_normal()
        bu[ms + dst] = bl[ms + dst] = val
        sense[ms + dst] = ACTIVE + IMMUTABLE
        if dep_eq and used < len(free):
            d2 = free[used]; used += 1
            A[d2] = 2.0 * A[dst]
            bu[ms + d2] = bl[ms + d2] = 2.0 * val * (1.0 if rng.random() < 0.7 else 1.3)
            sense[ms + d2] = ACTIVE + IMMUTABLE
    for _ in range(n_soft):
        if used >= len(free):
            break
        dst = free[used]; used += 1
        sense[ms + dst] = SOFT
        bu[ms + dst] = A[dst] @ x - 0.2 * rng.random()
        bl[ms + dst] = bu[ms + dst] - 1.0
    q.update(A=A, bupper=bu, blower=bl, sense=sense)
    return q


def generate_singular_qp(n, m, ms, rank, rng, kind="dense", in_range=False):
    """A feasible QP whose Hessian is only positive SEMI-definite (the proximal outer loop's input, daqp_prox.c):
    kind 'dense': H = T'T with T rank x n; 'diag': a diagonal H with zeros in ~40% of its coordinates.
    Constraints are random rows around a random interior point, so the QP is bounded and strictly feasible.
    in_range: f = H g, so the minimiser is not a vertex and not unique -- the proximal iteration then creeps towards
    the solution nearest to its start over many outer iterations (relaxation and confirmation steps included)."""
    rng = np.random.default_rng(rng)
    if kind == "diag":
        d = rng.random(n) + 0.5
        d[rng.random(n) < 0.4] = 0.0
        H = np.diag(d)
    else:
        T = rng.standard_normal((rank, n))
        H = T.T @ T
    f = rng.standard_normal(n)
    if in_range:
        f = H @ f
    A = rng.standard_normal((m - ms, n))
    x0 = rng.standard_normal(n)
    s = np.concatenate([x0[:ms], A @ x0])
    bu = s + 0.1 + rng.random(m)
    bl = s - 0.1 - rng.random(m)
    return dict(H=H, f=f, A=A, bupper=bu, blower=bl, sense=np.zeros(m, np.int32))


def generate_lp(n, m, ms, rng, unbounded=False):
    """min f'x over a random polytope around a random point (H is None: the reference's LP branch of daqp_prox.c).
    The first ms rows are simple bounds.  unbounded: most rows are dropped to one-sided so that -f is a recession direction."""
    rng = np.random.default_rng(rng)
    f = rng.standard_normal(n)
    A = rng.standard_normal((m - ms, n))
    x0 = rng.standard_normal(n)
    s = np.concatenate([x0[:ms], A @ x0])
    bu = s + 0.1 + rng.random(m)
    bl = s - 0.1 - rng.random(m)
    if unbounded:
        d = np.concatenate([-f[:ms], A @ (-f)])
        bu[d > 0] = 1e30
        bl[d < 0] = -1e30
    return dict(H=None, f=f, A=A, bupper=bu, blower=bl, sense=np.zeros(m, np.int32))


def generate_equality_qp(n, m, ms, neq, rng):
    """A generator QP with neq of its general rows turned into equalities (sense 5) through a common point: with
    neq > 5 and 10 neq > n the reference's daqp_quadprog eliminates them first (eq_elim.c:127-164)."""
    q = generate_qp(n, m, ms, max(1, min(n // 3, m - ms - neq - 1)), rng=rng)
    q = {k: q[k] for k in ("H", "f", "A", "bupper", "blower", "sense")}
    r = np.random.default_rng(rng)
    x0 = 0.1 * r.standard_normal(n)
    idx = ms + r.choice(m - ms, neq, replace=False)
    full = np.concatenate([x0[:ms], q["A"] @ x0])
    bu = np.maximum(q["bupper"], full + 0.05)
    bl = np.minimum(q["blower"], full - 0.05)
    s = q["sense"].copy()
    for j in idx:
        bu[j] = bl[j] = full[j]
        s[j] = 5
    q.update(bupper=bu, blower=bl, sense=s)
    return q


def add_sense_variety(q, ms, n_eq, n_soft, rng):
    """Turn n_eq general rows of a generated problem into equalities through the midpoint of their bounds (sense 5) and
    flag n_soft others as soft (sense 8): the equality / soft-constraint branches inside the proximal loop."""
    r = np.random.default_rng(rng)
    m = q["bupper"].size
    rows = ms + r.permutation(m - ms)
    s = q["sense"].copy()
    bu, bl = q["bupper"].copy(), q["blower"].copy()
    for j in rows[:n_eq]:
        mid = 0.5 * (min(bu[j], 1e3) + max(bl[j], -1e3))
        bu[j] = bl[j] = mid
        s[j] = 5
    for j in rows[n_eq:n_eq + n_soft]:
        s[j] = 8
    out = dict(q)
    out.update(bupper=bu, blower=bl, sense=s)
    return out
