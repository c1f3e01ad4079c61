"""Host-side mirror of the reference's Python binding for the dense QP path, plus its batch twin.

  solve(H, f, A, bupper, blower, sense, **settings)      reference daqp.pyx:68-221  -> daqp_quadprog
  Model().setup/solve/update                              reference daqp.pyx:222-572 -> setup_daqp /
                                                          daqp_solve / daqp_update_ldp
  solve_batch(...), BatchModel                            N problems of one shape -> daqp_batch_*

Everything goes through the C ABI of libdaqp_amd.so; torch tensors on the GPU are passed by
device pointer (no copy), numpy arrays are staged by the library.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (DAQPBatchProblem, DAQPBatchResult, DAQPProblem, DAQPResult, MEM_DEVICE, MEM_HOST,
                   c_double_p, c_int_p, default_settings, lib)

UPDATE_Rinv, UPDATE_M, UPDATE_v, UPDATE_d, UPDATE_sense = 1, 2, 4, 8, 16
UPDATE_unconstrained, UPDATE_eliminate = 64, 128
INF = 1e30
TRACE_MARK = 0x40000000
TRACE_PIVOT, TRACE_SINGULAR, TRACE_REFINE, TRACE_REFACTOR, TRACE_CYCLE_RESET = (TRACE_MARK + k for k in range(1, 6))

try:  # torch is plumbing only (device memory, streams); the package works on numpy without it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _np64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _np32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_int_p)


# ---------------------------------------------------------------------------
# single problem: same call shape as the reference binding
# ---------------------------------------------------------------------------
def solve(H, f, A, bupper, blower=None, sense=None, **settings):
    """x, fval, exitflag, info = solve(H, f, A, bupper, blower, sense, primal_tol=..., iter_limit=...)"""
    H, f, A, bupper = _np64(H), _np64(f), _np64(A), _np64(bupper)
    n, m = f.size, bupper.size
    mA = A.shape[0] if (A is not None and A.ndim == 2) else 0
    blower = np.full(m, -INF) if blower is None else _np64(blower)
    sense = np.zeros(m, np.int32) if sense is None else _np32(sense)
    x, lam = np.empty(n), np.empty(m)
    qp = DAQPProblem(n, m, m - mA, _dp(H), _dp(f), _dp(A) if mA else None, _dp(bupper), _dp(blower), _ip(sense),
                     None, 0, 0)
    st = default_settings(**settings)
    res = DAQPResult(_dp(x), _dp(lam), 0, 0, 0, 0, 0, 0, 0)
    lib().daqp_quadprog(C.byref(res), C.byref(qp), C.byref(st))
    if res.exitflag == -8 and _lib.last_error():
        info_err = _lib.last_error()
    else:
        info_err = ""
    return x, res.fval, res.exitflag, {"solve_time": res.solve_time, "setup_time": res.setup_time,
                                       "iterations": res.iter, "nodes": res.nodes, "lam": lam, "error": info_err}


class Model:
    """Persistent workspace (reference daqp.pyx:222-572): setup once, then solve / update / solve ..."""

    def __init__(self):
        self._ws = None
        self._keep = {}
        self._qp = None
        self._settings = default_settings()
        self.n = self.m = self.ms = 0

    def _problem(self):
        k = self._keep
        self._qp = DAQPProblem(self.n, self.m, self.ms, _dp(k["H"]), _dp(k["f"]), _dp(k.get("A")), _dp(k["bupper"]),
                               _dp(k["blower"]), _ip(k.get("sense")), None, 0, 0)
        return self._qp

    def setup(self, H, f, A, bupper, blower=None, sense=None, **settings):
        self._free()
        H, f, A, bupper = _np64(H), _np64(f), _np64(A), _np64(bupper)
        self.n, self.m = f.size, bupper.size
        mA = A.shape[0] if (A is not None and A.ndim == 2) else 0
        self.ms = self.m - mA
        blower = np.full(self.m, -INF) if blower is None else _np64(blower)
        sense = np.zeros(self.m, np.int32) if sense is None else _np32(sense)
        self._keep = dict(H=H, f=f, A=A if mA else None, bupper=bupper, blower=blower, sense=sense)
        self._settings = default_settings(**settings)
        self._ws = C.create_string_buffer(_lib.WORKSPACE_BYTES)
        C.c_void_p.from_buffer(self._ws, 224).value = C.addressof(self._settings)  # work->settings (borrowed)
        t = C.c_double(0)
        flag = lib().setup_daqp(C.byref(self._problem()), self._ws, C.byref(t))
        if flag < 0:
            self._ws = None
        return flag, t.value

    def settings(self, **kw):
        for k, v in kw.items():
            setattr(self._settings, k, v)
        return {k: getattr(self._settings, k) for k, _ in self._settings._fields_}

    def update(self, H=None, f=None, A=None, bupper=None, blower=None, sense=None):
        """daqp.pyx:513-571: the mask is built field by field -- one bit per array given, arrays of the wrong shape are ignored as
        the reference ignores them -- and handed to daqp_update_ldp as it is (utils.c:58-221 runs each bit's step on its own: a
        new A keeps the factor, a new sense keeps everything else, ...)."""
        if self._ws is None:
            raise RuntimeError("Model.update called before setup")
        n, m, mA = self.n, self.m, self.m - self.ms
        mask = 0
        if H is not None and np.shape(H) == (n, n):
            self._keep["H"] = _np64(H); mask |= UPDATE_Rinv
        if A is not None and np.shape(A) == (mA, n):
            self._keep["A"] = _np64(A); mask |= UPDATE_M
        if f is not None and np.size(f) == n:
            self._keep["f"] = _np64(f); mask |= UPDATE_v
        if bupper is not None and np.size(bupper) == m:
            self._keep["bupper"] = _np64(bupper); mask |= UPDATE_d
        if blower is not None and np.size(blower) == m:
            self._keep["blower"] = _np64(blower); mask |= UPDATE_d
        if sense is not None and np.size(sense) == m:
            self._keep["sense"] = _np32(sense); mask |= UPDATE_sense
        return lib().daqp_update_ldp(mask, self._ws, C.byref(self._problem()))

    def update_mask(self, mask, **arrays):
        """daqp_update_ldp(mask, ...) with an explicit mask (C callers' usage): `arrays` (H, f, A, bupper, blower, sense) replace
        what the problem descriptor points to, whatever the mask says."""
        if self._ws is None:
            raise RuntimeError("Model.update_mask called before setup")
        for k, v in arrays.items():
            if v is not None:
                self._keep[k] = _np32(v) if k == "sense" else _np64(v)
        return lib().daqp_update_ldp(int(mask), self._ws, C.byref(self._problem()))

    def solve(self):
        if self._ws is None:
            raise RuntimeError("Model.solve called before setup")
        x, lam = np.empty(self.n), np.empty(self.m)
        res = DAQPResult(_dp(x), _dp(lam), 0, 0, 0, 0, 0, 0, 0)
        lib().daqp_solve(C.byref(res), self._ws)
        return x, res.fval, res.exitflag, {"solve_time": res.solve_time, "setup_time": 0.0, "iterations": res.iter,
                                           "nodes": res.nodes, "lam": lam}

    def _free(self):
        if self._ws is not None:
            C.c_void_p.from_buffer(self._ws, 224).value = None  # borrowed settings are not ours to free
            lib().free_daqp_workspace(self._ws)
            lib().free_daqp_ldp(self._ws)
            self._ws = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# batches
# ---------------------------------------------------------------------------
def _is_torch(a):
    return torch is not None and isinstance(a, torch.Tensor)


def _ptr(a, dtype, keep):
    """(pointer, memory) of a numpy array or torch tensor, made contiguous and of the right dtype."""
    if a is None:
        return None, None
    if _is_torch(a):
        tdt = torch.float64 if dtype == np.float64 else torch.int32
        t = a.to(tdt).contiguous()
        keep.append(t)
        return t.data_ptr(), (MEM_DEVICE if t.is_cuda else MEM_HOST)
    arr = np.ascontiguousarray(a, dtype=dtype)
    keep.append(arr)
    return arr.ctypes.data, MEM_HOST


class BatchModel:
    """Device-resident workspaces of N QPs of one shape: setup -> solve -> {update(f, bounds) -> solve}*.

    Inputs may be numpy arrays (staged over PCIe by the library) or CUDA/HIP torch tensors (used in
    place; they must stay alive until the next setup/update replaces them).  Shapes: H (N,n,n),
    f (N,n), A (N,m-ms,n), bupper/blower (N,m), sense (N,m) int32 or None.
    """

    def __init__(self, N, n, m, ms=0, ns_max=0, device=None, **settings):
        self.N, self.n, self.m, self.ms, self.ns = N, n, m, ms, ns_max
        self._settings = default_settings(**settings)
        h = C.c_void_p()
        dev = -1 if device is None else int(device)
        rc = lib().daqp_batch_create(C.byref(h), N, n, m, ms, ns_max, C.byref(self._settings), dev)
        if rc != 0:
            raise RuntimeError(f"daqp_batch_create failed ({rc}): {_lib.last_error()}")
        self._h = h
        self._keep = {}          # name -> array/tensor the device currently reads (one entry per input, replaced one by one)
        self._out_keep = []
        if torch is not None and torch.cuda.is_available():
            # the batch lives on ONE device: its launches go to that device's current stream and its outputs are allocated there
            self.device = torch.cuda.current_device() if device is None else int(device)
            lib().daqp_batch_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        else:
            self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().daqp_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _problem(self, H=None, f=None, A=None, bupper=None, blower=None, sense=None):
        """DAQPBatchProblem of the given arrays.  Device arrays are used in place by the library (and bounds / f are read
        again by every later update and solve), so each one stays referenced under its own name until a later call
        replaces THAT array -- an update(f=...) must not drop the bounds adopted at setup."""
        keep = {}
        mems = set()
        ptrs = []
        for name, a, dt in (("H", H, np.float64), ("f", f, np.float64), ("A", A, np.float64), ("bupper", bupper, np.float64),
                            ("blower", blower, np.float64), ("sense", sense, np.int32)):
            tmp = []
            p, mem = _ptr(a, dt, tmp)
            ptrs.append(p)
            if mem is not None:
                mems.add(mem)
                keep[name] = tmp[0]
        if len(mems) > 1:
            raise ValueError("mix of host and device arrays in one call")
        mem = mems.pop() if mems else MEM_HOST
        if mem == MEM_DEVICE:
            self._keep.update(keep)
        return DAQPBatchProblem(self.N, self.n, self.m, self.ms, *ptrs, mem), keep

    def setup(self, H, f, A, bupper, blower, sense=None, init_mask=0):
        p, keep = self._problem(H, f, A, bupper, blower, sense)
        rc = lib().daqp_batch_setup(self._h, C.byref(p), init_mask)
        if rc != 0:
            raise RuntimeError(f"daqp_batch_setup failed ({rc}): {_lib.last_error()}")
        return self

    def setup_shared(self, H, f, A, bupper, blower, sense=None):
        """N problems with ONE H (n, n) and ONE A (m-ms, n) -- condensed MPC: same plant, per-problem f / bounds.
        Same result as Model.setup once (open bounds) followed by Model.update(f_k, bu_k, bl_k) per problem -- the reference's
        MPC usage; the factorisation runs once."""
        assert H.ndim == 2 and (A is None or A.ndim == 2), "setup_shared takes a single H and a single A"
        p, keep = self._problem(H, f, A, bupper, blower, sense)
        rc = lib().daqp_batch_setup_shared(self._h, C.byref(p), 0)
        if rc != 0:
            raise RuntimeError(f"daqp_batch_setup_shared failed ({rc}): {_lib.last_error()}")
        return self

    def setup_flags(self):
        fl = np.zeros(self.N, np.int32)
        lib().daqp_batch_setup_flags(self._h, _ip(fl))
        return fl

    def update(self, f=None, bupper=None, blower=None, H=None, A=None, sense=None, mask=None):
        """daqp_update_ldp for every problem, the mask built field by field like Model.update (daqp.pyx:513-571): f -> v, bounds -> d
        (factors and working sets kept: the warm path), A -> M on the kept factor, H -> a new factor (v, M, d follow), sense ->
        the working sets are rebuilt from its ACTIVE bits.  mask: an explicit DAQP_UPDATE_* mask instead (sense=None with the
        sense bit means "all zeros", utils.c:85-86)."""
        if mask is None:
            mask = ((UPDATE_Rinv if H is not None else 0) | (UPDATE_M if A is not None else 0) | (UPDATE_v if f is not None else 0)
                    | (UPDATE_d if (bupper is not None or blower is not None) else 0) | (UPDATE_sense if sense is not None else 0))
        p, keep = self._problem(H, f, A, bupper, blower, sense)
        rc = lib().daqp_batch_update(self._h, int(mask), C.byref(p))
        if rc != 0:
            raise RuntimeError(f"daqp_batch_update failed ({rc}): {_lib.last_error()}")
        return self

    def solve(self, out="numpy"):
        """Returns dict(x, lam, fval, exitflag, iter, soft_slack); out='numpy' or 'torch' (device tensors)."""
        N, n, m = self.N, self.n, self.m
        if out == "torch":
            dev = torch.device("cuda", self.device)
            o = dict(x=torch.empty((N, n), dtype=torch.float64, device=dev),
                     lam=torch.empty((N, m), dtype=torch.float64, device=dev),
                     fval=torch.empty(N, dtype=torch.float64, device=dev),
                     soft_slack=torch.empty(N, dtype=torch.float64, device=dev),
                     exitflag=torch.empty(N, dtype=torch.int32, device=dev),
                     iter=torch.empty(N, dtype=torch.int32, device=dev))
            r = DAQPBatchResult(o["x"].data_ptr(), o["lam"].data_ptr(), o["fval"].data_ptr(), o["soft_slack"].data_ptr(),
                                o["exitflag"].data_ptr(), o["iter"].data_ptr(), MEM_DEVICE, 0, 0)
        else:
            o = dict(x=np.empty((N, n)), lam=np.empty((N, m)), fval=np.empty(N), soft_slack=np.empty(N),
                     exitflag=np.empty(N, np.int32), iter=np.empty(N, np.int32))
            r = DAQPBatchResult(o["x"].ctypes.data, o["lam"].ctypes.data, o["fval"].ctypes.data, o["soft_slack"].ctypes.data,
                                o["exitflag"].ctypes.data, o["iter"].ctypes.data, MEM_HOST, 0, 0)
        rc = lib().daqp_batch_solve(self._h, C.byref(r))
        if rc != 0:
            raise RuntimeError(f"daqp_batch_solve failed ({rc}): {_lib.last_error()}")
        self._out_keep = [o]
        return o

    def set_primal_start(self, x):
        """api.c:636-641 for every problem: where the proximal iterations (singular H) start from; x: (N, n)."""
        keep = []
        ptr, mem = _ptr(x, np.float64, keep)
        rc = lib().daqp_batch_set_primal_start(self._h, ptr, mem)
        if rc != 0:
            raise RuntimeError(f"daqp_batch_set_primal_start failed ({rc}): {_lib.last_error()}")
        return self

    def prox_info(self):
        """dict(n_prox, outer, eps): which problems go through the proximal outer loop (singular H), the outer
        iterations of the last solve and the shift of each factor."""
        npx, outer, eps = np.zeros(self.N, np.int32), np.zeros(self.N, np.int32), np.zeros(self.N)
        lib().daqp_batch_prox_info(self._h, _ip(npx), _ip(outer), eps.ctypes.data)
        return dict(n_prox=npx, outer=outer, eps=eps)

    def kernel_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        lib().daqp_batch_kernel_ms(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def working_sets(self):
        cap = self.n + self.ns + 1
        na, ws = np.zeros(self.N, np.int32), np.zeros((self.N, cap), np.int32)
        lib().daqp_batch_working_sets(self._h, _ip(na), _ip(ws))
        return na, ws

    def device_bytes(self):
        return int(lib().daqp_batch_device_bytes(self._h))

    def rechecked(self):
        """problems of the last solve whose INFEASIBLE verdict was re-derived in the reference's arithmetic (default mode, first
        solve after a setup: include/daqp_amd.h, daqp_batch_rechecked)"""
        return int(lib().daqp_batch_rechecked(self._h))

    def set_recheck(self, on):
        """switch the second pass of INFEASIBLE verdicts on / off for this batch (daqp_batch_set_recheck: off for a caller that
        recycles device-resident input buffers between setup and solve)"""
        lib().daqp_batch_set_recheck(self._h, 1 if on else 0)

    def recheck_ms(self):
        t = C.c_float(0)
        lib().daqp_batch_recheck_ms(self._h, C.byref(t))
        return t.value

    # test hooks
    def enable_trace(self, cap=4096):
        lib().daqp_batch_enable_trace(self._h, cap)
        self._trace_cap = cap

    def read_trace(self, marks=False):
        """Per problem: +(id+1) for an added constraint, -(id+1) for a removed one, in order.  marks=True keeps the branch
        markers as well (TRACE_PIVOT, TRACE_SINGULAR, TRACE_REFINE, TRACE_REFACTOR, TRACE_CYCLE_RESET)."""
        t = np.zeros((self.N, self._trace_cap), np.int32)
        lib().daqp_batch_read_trace(self._h, _ip(t))
        out = [t[q, : min(t[q, -1], self._trace_cap - 1)].copy() for q in range(self.N)]
        return out if marks else [e[e < TRACE_MARK] for e in out]

    def enable_profile(self, on=True):
        lib().daqp_batch_enable_profile(self._h, 1 if on else 0)

    def read_profile(self):
        """(N, 32) int64 per QP.  Solve kernel (register variant): cycles per state machine state [0:16] and
        visits [16:32]; setup kernel: cycles per phase [0:6]."""
        p = np.zeros((self.N, 32), np.int64)
        lib().daqp_batch_read_profile(self._h, p.ctypes.data_as(C.POINTER(C.c_longlong)))
        return p

    def read_ldp(self, q):
        n, m, ms = self.n, self.m, self.ms
        M, R, v = np.zeros((m - ms, n)), np.zeros(n * (n + 1) // 2), np.zeros(n)
        du, dl, sc = np.zeros(m), np.zeros(m), np.zeros(m)
        lib().daqp_batch_read_ldp(self._h, q, _dp(M), _dp(R), _dp(v), _dp(du), _dp(dl), _dp(sc))
        return M, R, v, du, dl, sc


def solve_batch(H, f, A, bupper, blower=None, sense=None, ms=None, out="numpy", **settings):
    """N x daqp_quadprog in one call (daqp_quadprog_batch semantics: unconstrained shortcut on).

    Returns dict(x, lam, fval, exitflag, iter, soft_slack)."""
    N, n = f.shape[0], f.shape[1]
    m = bupper.shape[1]
    mA = A.shape[1] if A is not None and A.ndim == 3 else 0
    ms = m - mA if ms is None else ms
    if blower is None:
        blower = (torch.full_like(bupper, -INF) if _is_torch(bupper) else np.full(bupper.shape, -INF))
    if N == 0:   # an empty batch is a valid (empty) answer, as it is for daqp_quadprog_batch
        if out == "torch" and _is_torch(f):
            z = lambda *sh, dt=torch.float64: torch.zeros(sh, dtype=dt, device=f.device)
            return dict(x=z(0, n), lam=z(0, m), fval=z(0), soft_slack=z(0), exitflag=z(0, dt=torch.int32), iter=z(0, dt=torch.int32))
        return dict(x=np.zeros((0, n)), lam=np.zeros((0, m)), fval=np.zeros(0), soft_slack=np.zeros(0),
                    exitflag=np.zeros(0, np.int32), iter=np.zeros(0, np.int32))
    ns = 0
    if sense is not None:
        s = sense.cpu().numpy() if _is_torch(sense) else np.asarray(sense)
        ns = int(((s & 8) != 0).sum(axis=1).max()) if s.size else 0
    bm = BatchModel(N, n, m, ms, ns, **settings)
    try:
        bm.setup(H, f, A, bupper, blower, sense, init_mask=UPDATE_unconstrained | UPDATE_eliminate)
        res = bm.solve(out=out)
        if out == "torch":
            torch.cuda.synchronize()
    finally:
        bm.close()
    return res


def solve_batch_multi(H, f, A, bupper, blower=None, sense=None, ms=None, devices=None, **settings):
    """solve_batch over several GPUs of this host: problem k on devices[k mod len(devices)] (daqp_quadprog_batch_multi: one host
    thread, stream and set of workspaces per shard, no exchange step).  numpy arrays in, numpy arrays out; devices=None: every
    visible device.  Returns dict(x, lam, fval, exitflag, iter, soft_slack)."""
    N, n = f.shape[0], f.shape[1]
    m = bupper.shape[1]
    mA = A.shape[1] if A is not None and A.ndim == 3 else 0
    ms = m - mA if ms is None else ms
    if blower is None:
        blower = np.full(bupper.shape, -INF)
    keep = []
    ptrs = [_ptr(a, dt, keep)[0] for a, dt in ((H, np.float64), (f, np.float64), (A if mA else None, np.float64), (bupper, np.float64),
                                                (blower, np.float64), (sense, np.int32))]
    p = DAQPBatchProblem(N, n, m, ms, *ptrs, MEM_HOST)
    o = dict(x=np.empty((N, n)), lam=np.empty((N, m)), fval=np.empty(N), soft_slack=np.empty(N),
             exitflag=np.empty(N, np.int32), iter=np.empty(N, np.int32))
    r = DAQPBatchResult(o["x"].ctypes.data, o["lam"].ctypes.data, o["fval"].ctypes.data, o["soft_slack"].ctypes.data,
                        o["exitflag"].ctypes.data, o["iter"].ctypes.data, MEM_HOST, 0, 0)
    st = default_settings(**settings)
    dev = None if devices is None else np.ascontiguousarray(devices, np.int32)
    rc = lib().daqp_quadprog_batch_multi(C.byref(r), C.byref(p), C.byref(st), _ip(dev), 0 if dev is None else dev.size)
    if rc != 0:
        raise RuntimeError(f"daqp_quadprog_batch_multi failed ({rc}): {_lib.last_error()}")
    return o


class MultiBatchModel:
    """N QPs of one shape held on several GPUs of this host between calls (include/daqp_amd.h, DAQPMultiBatch): problem k lives on
    devices[k mod G].  setup -> solve -> {update(f, bounds) -> solve}* with ONE host-resident batch in the caller's order, numpy in,
    numpy out -- BatchModel's sequence, sharded.  devices=None: every visible device; a device may be listed more than once."""

    def __init__(self, N, n, m, ms=0, ns_max=0, devices=None, **settings):
        self.N, self.n, self.m, self.ms = N, n, m, ms
        self._settings = default_settings(**settings)
        dev = None if devices is None else np.ascontiguousarray(devices, np.int32)
        h = C.c_void_p()
        rc = lib().daqp_batch_create_multi(C.byref(h), N, n, m, ms, ns_max, C.byref(self._settings), _ip(dev), 0 if dev is None else dev.size)
        if rc != 0:
            raise RuntimeError(f"daqp_batch_create_multi failed ({rc}): {_lib.last_error()}")
        self._h = h
        self.shards = int(lib().daqp_batch_multi_shards(h))

    def close(self):
        if getattr(self, "_h", None):
            lib().daqp_batch_free_multi(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _problem(self, keep, H=None, f=None, A=None, bupper=None, blower=None, sense=None):
        mA = self.m - self.ms
        ptrs = [_ptr(a, dt, keep)[0] for a, dt in ((H, np.float64), (f, np.float64), (A if mA else None, np.float64), (bupper, np.float64),
                                                    (blower, np.float64), (sense, np.int32))]
        return DAQPBatchProblem(self.N, self.n, self.m, self.ms, *ptrs, MEM_HOST)

    def setup(self, H, f, A, bupper, blower=None, sense=None, init_mask=0):
        if blower is None:
            blower = np.full(np.shape(bupper), -INF)
        keep = []
        p = self._problem(keep, H, f, A, bupper, blower, sense)
        rc = lib().daqp_batch_setup_multi(self._h, C.byref(p), init_mask)
        if rc != 0:
            raise RuntimeError(f"daqp_batch_setup_multi failed ({rc}): {_lib.last_error()}")

    def update(self, f=None, bupper=None, blower=None, H=None, A=None, sense=None, mask=None):
        """BatchModel.update, sharded: any mask of daqp_update_ldp (built field by field unless given)"""
        if mask is None:
            mask = ((UPDATE_Rinv if H is not None else 0) | (UPDATE_M if A is not None else 0) | (UPDATE_v if f is not None else 0)
                    | (UPDATE_d if (bupper is not None or blower is not None) else 0) | (UPDATE_sense if sense is not None else 0))
        keep = []
        p = self._problem(keep, H, f, A, bupper, blower, sense)
        rc = lib().daqp_batch_update_multi(self._h, int(mask), C.byref(p))
        if rc != 0:
            raise RuntimeError(f"daqp_batch_update_multi failed ({rc}): {_lib.last_error()}")

    def solve(self):
        N, n, m = self.N, self.n, self.m
        o = dict(x=np.empty((N, n)), lam=np.empty((N, m)), fval=np.empty(N), soft_slack=np.empty(N),
                 exitflag=np.empty(N, np.int32), iter=np.empty(N, np.int32))
        r = DAQPBatchResult(o["x"].ctypes.data, o["lam"].ctypes.data, o["fval"].ctypes.data, o["soft_slack"].ctypes.data,
                            o["exitflag"].ctypes.data, o["iter"].ctypes.data, MEM_HOST, 0, 0)
        rc = lib().daqp_batch_solve_multi(self._h, C.byref(r))
        if rc != 0:
            raise RuntimeError(f"daqp_batch_solve_multi failed ({rc}): {_lib.last_error()}")
        o["solve_time"] = r.solve_time
        return o

    # ---- per-shard descriptors (daqp_batch_*_multi_shards): shard g's problems back to back, host arrays or tensors ON shard g's device
    def _shard_problems(self, shards, names):
        P = (DAQPBatchProblem * self.shards)()
        keep = []
        for g, sh in enumerate(shards):
            _, sn, dev = self.shard(g)
            ptrs, mems = [], set()
            for name in ("H", "f", "A", "bupper", "blower", "sense"):
                a = sh.get(name) if name in names else None
                if name == "A" and self.m == self.ms:
                    a = None
                ptr, mem = _ptr(a, np.int32 if name == "sense" else np.float64, keep)
                ptrs.append(ptr)
                if mem is not None:
                    mems.add(mem)
            if len(mems) > 1:
                raise ValueError("mix of host and device arrays in one shard")
            P[g] = DAQPBatchProblem(sn, self.n, self.m, self.ms, *ptrs, mems.pop() if mems else MEM_HOST)
        return P, keep

    def setup_shards(self, shards, init_mask=0):
        """shards[g]: dict(H, f, A, bupper, blower[, sense]) of shard g's problems (problem k of the batch is problem k // G of shard
        k mod G), numpy or torch tensors resident on that shard's device (used in place)"""
        P, keep = self._shard_problems(shards, ("H", "f", "A", "bupper", "blower", "sense"))
        self._shard_keep = keep
        rc = lib().daqp_batch_setup_multi_shards(self._h, P, init_mask)
        if rc != 0:
            raise RuntimeError(f"daqp_batch_setup_multi_shards failed ({rc}): {_lib.last_error()}")

    def update_shards(self, shards, mask):
        P, keep = self._shard_problems(shards, ("H", "f", "A", "bupper", "blower", "sense"))
        self._shard_keep_upd = keep
        rc = lib().daqp_batch_update_multi_shards(self._h, int(mask), P)
        if rc != 0:
            raise RuntimeError(f"daqp_batch_update_multi_shards failed ({rc}): {_lib.last_error()}")

    def solve_shards(self, out="torch"):
        """one result dict per shard; out='torch': tensors on the shard's device, 'numpy': host arrays"""
        R = (DAQPBatchResult * self.shards)()
        outs = []
        for g in range(self.shards):
            _, sn, dev = self.shard(g)
            if out == "torch":
                d = torch.device("cuda", dev)
                o = dict(x=torch.empty((sn, self.n), dtype=torch.float64, device=d), lam=torch.empty((sn, self.m), dtype=torch.float64, device=d),
                         fval=torch.empty(sn, dtype=torch.float64, device=d), soft_slack=torch.empty(sn, dtype=torch.float64, device=d),
                         exitflag=torch.empty(sn, dtype=torch.int32, device=d), iter=torch.empty(sn, dtype=torch.int32, device=d))
                R[g] = DAQPBatchResult(o["x"].data_ptr(), o["lam"].data_ptr(), o["fval"].data_ptr(), o["soft_slack"].data_ptr(),
                                       o["exitflag"].data_ptr(), o["iter"].data_ptr(), MEM_DEVICE, 0, 0)
            else:
                o = dict(x=np.empty((sn, self.n)), lam=np.empty((sn, self.m)), fval=np.empty(sn), soft_slack=np.empty(sn),
                         exitflag=np.empty(sn, np.int32), iter=np.empty(sn, np.int32))
                R[g] = DAQPBatchResult(o["x"].ctypes.data, o["lam"].ctypes.data, o["fval"].ctypes.data, o["soft_slack"].ctypes.data,
                                       o["exitflag"].ctypes.data, o["iter"].ctypes.data, MEM_HOST, 0, 0)
            outs.append(o)
        rc = lib().daqp_batch_solve_multi_shards(self._h, R)
        if rc != 0:
            raise RuntimeError(f"daqp_batch_solve_multi_shards failed ({rc}): {_lib.last_error()}")
        return outs

    def shard_kernel_ms(self, g=0):
        a, b = C.c_float(0), C.c_float(0)
        lib().daqp_batch_kernel_ms(self.shard(g)[0], C.byref(a), C.byref(b))
        return a.value, b.value

    def shard(self, g):
        """(DAQPBatch handle, problems on it, device) of shard g -- for the single-device inspection calls"""
        sn, dev = C.c_int(0), C.c_int(0)
        h = lib().daqp_batch_multi_shard(self._h, g, C.byref(sn), C.byref(dev))
        return h, sn.value, dev.value
