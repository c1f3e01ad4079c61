"""Loader/builder for libdaqp_amd.so (the C ABI of include/daqp_amd.h).

The shared library is compiled in-tree with hipcc for gfx950 and loaded with ctypes.  There is
no Python or CPU implementation of the solver behind it: if the library cannot be built or
loaded, importing the solver entry points raises.
"""
import ctypes as C
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIBPATH = os.path.join(LIBDIR, "libdaqp_amd.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class DAQPSettings(C.Structure):  # include/daqp_amd.h (layout of reference types.h:52-74)
    _fields_ = [("primal_tol", C.c_double), ("dual_tol", C.c_double), ("zero_tol", C.c_double),
                ("pivot_tol", C.c_double), ("progress_tol", C.c_double),
                ("cycle_tol", C.c_int), ("iter_limit", C.c_int),
                ("fval_bound", C.c_double), ("eps_prox", C.c_double), ("eta_prox", C.c_double),
                ("rho_soft", C.c_double), ("rel_subopt", C.c_double), ("abs_subopt", C.c_double),
                ("sing_tol", C.c_double), ("refactor_tol", C.c_double), ("time_limit", C.c_double)]


class DAQPProblem(C.Structure):  # reference types.h:14-50
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("ms", C.c_int),
                ("H", c_double_p), ("f", c_double_p), ("A", c_double_p),
                ("bupper", c_double_p), ("blower", c_double_p), ("sense", c_int_p),
                ("break_points", c_int_p), ("nh", C.c_int), ("problem_type", C.c_int)]


class DAQPResult(C.Structure):  # reference api.h:15-27
    _fields_ = [("x", c_double_p), ("lam", c_double_p), ("fval", C.c_double), ("soft_slack", C.c_double),
                ("exitflag", C.c_int), ("iter", C.c_int), ("nodes", C.c_int),
                ("solve_time", C.c_double), ("setup_time", C.c_double)]


class DAQPBatchProblem(C.Structure):
    _fields_ = [("N", C.c_int), ("n", C.c_int), ("m", C.c_int), ("ms", C.c_int),
                ("H", C.c_void_p), ("f", C.c_void_p), ("A", C.c_void_p),
                ("bupper", C.c_void_p), ("blower", C.c_void_p), ("sense", C.c_void_p),
                ("memory", C.c_int)]


class DAQPBatchResult(C.Structure):
    _fields_ = [("x", C.c_void_p), ("lam", C.c_void_p), ("fval", C.c_void_p), ("soft_slack", C.c_void_p),
                ("exitflag", C.c_void_p), ("iter", C.c_void_p), ("memory", C.c_int),
                ("setup_time", C.c_double), ("solve_time", C.c_double)]


WORKSPACE_BYTES = 288  # sizeof(DAQPWorkspace) in include/daqp_amd.h == reference types.h:187-264
MEM_HOST, MEM_DEVICE = 0, 1

EXPORTS = [
    "daqp_quadprog", "daqp_solve", "setup_daqp", "setup_daqp_main", "daqp_update_ldp", "daqp_default_settings",
    "allocate_daqp_settings", "free_daqp_workspace", "free_daqp_ldp", "daqp_primal_init_active",
    "daqp_dual_init_active", "daqp_set_primal_start", "daqp_minrep", "allocate_daqp_workspace", "allocate_daqp_ldp", "daqp_first_violating", "daqp_batch_create", "daqp_batch_free", "daqp_amd_release_pool", "daqp_amd_shutdown", "daqp_batch_set_stream",
    "daqp_batch_set_settings", "daqp_batch_set_exact", "daqp_batch_setup", "daqp_batch_setup_shared", "daqp_batch_update", "daqp_batch_solve",
    "daqp_batch_setup_flags", "daqp_batch_working_sets", "daqp_batch_set_primal_start", "daqp_batch_prox_info", "daqp_quadprog_batch", "daqp_quadprog_batch_multi", "daqp_batch_kernel_ms",
    "daqp_batch_create_multi", "daqp_batch_free_multi", "daqp_batch_multi_shards", "daqp_batch_multi_shard", "daqp_batch_setup_multi", "daqp_batch_update_multi",
    "daqp_batch_solve_multi", "daqp_batch_setup_multi_shards", "daqp_batch_update_multi_shards", "daqp_batch_solve_multi_shards",
    "daqp_batch_device_bytes", "daqp_batch_rechecked", "daqp_batch_set_recheck", "daqp_batch_recheck_ms", "daqp_amd_last_error", "daqp_amd_device_count", "daqp_amd_version", "daqp_amd_has_tiny",
    "setup_daqp_ldp", "daqp_ldp", "ldp2qp_solution", "daqp_extract_result",
    "daqp_batch_enable_trace", "daqp_batch_read_trace", "daqp_batch_enable_profile", "daqp_batch_read_profile", "daqp_batch_read_ldp",
]


# translation units -> what each one includes (a unit is recompiled when any of these is newer than its object, or when the
# object was compiled with other flags)
UNITS = {
    "daqp_amd.hip": ["daqp_amd.hip", "kernels.hip.h", "wave_ldp.hip.h", "wave_ldp_reg.hip.h", "setup_fast.hip.h", "prox.hip.h",
                     "wg_layout.hip.h", "batch_dev.hip.h", "recheck.hip.h", "setup_m.hip.h", "setup_fact.hip.h", "reg_kernel.hip.h", "tiny_setup.hip.h", "setup_blk.hip.h", "multi.hip.h"],
    "reg_kernel.hip": ["reg_kernel.hip", "reg_kernel.hip.h", "wave_ldp_reg.hip.h", "wave_ldp.hip.h", "batch_dev.hip.h"],
    "reg32_kernel.hip": ["reg32_kernel.hip", "reg_kernel.hip.h", "wave_ldp_reg.hip.h", "wave_ldp.hip.h", "batch_dev.hip.h"],
    "wg_kernel.hip": ["wg_kernel.hip", "wg_kernel.hip.h", "wg_ldp.hip.h", "wg_layout.hip.h", "batch_dev.hip.h", "wave_ldp.hip.h", "wave_ldp_reg.hip.h"],
    "setup_kernel.hip": ["setup_kernel.hip", "setup_blk.hip.h", "setup_fast.hip.h", "setup_m.hip.h", "setup_fact.hip.h", "tiny_setup.hip.h", "wave_ldp_reg.hip.h", "wave_ldp.hip.h", "batch_dev.hip.h"],
}
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
OBJDIR = os.path.join(LIBDIR, "obj")
UNITS = {u: d for u, d in UNITS.items() if os.path.exists(os.path.join(CSRC, u))}


def units(extra_flags=()):
    """translation units of a build with these flags"""
    return list(UNITS)


def _flag_key(extra_flags=()):
    return " ".join([*HIPFLAGS, *extra_flags])


def _unit_stale(unit, extra_flags=()):
    obj = os.path.join(OBJDIR, unit + ".o")
    if not os.path.exists(obj) or not os.path.exists(obj + ".resources.txt"):
        return True
    try:   # an object compiled with other flags (a development build's -DDAQP_AMD_FEW_VARIANTS, say) is not this build's object
        with open(obj + ".flags") as fh:
            if fh.read() != _flag_key(extra_flags):
                return True
    except OSError:
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, d) for d in UNITS[unit]] + [os.path.join(ROOT, "include", "daqp_amd.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _stale(extra_flags=()):
    if not os.path.exists(LIBPATH):
        return True
    if any(_unit_stale(u, extra_flags) or os.path.getmtime(os.path.join(OBJDIR, u + ".o")) > os.path.getmtime(LIBPATH) for u in units(extra_flags)):
        return True
    if extra_flags:
        return False
    try:   # a development build (tools/devbuild.sh: fewer kernel variants) is never what build() should leave behind
        L = C.CDLL(LIBPATH)
        v = L.daqp_amd_version
        v.restype = C.c_char_p
        return b"dev build" in v()
    except (OSError, AttributeError):
        return True


def build(force=False, verbose=False, extra_flags=()):
    """hipcc --offload-arch=gfx950 -> daqp_amd/lib/libdaqp_amd.so (cross-compiles without a GPU).  One object per translation
    unit (compiled side by side, only the stale ones: older than a source, or compiled with other flags), then one link.
    Compile and link run under an exclusive file lock and the library is moved into place atomically, so the eight ranks of
    a torchrun launch that import a stale tree build it once, not eight times on top of each other."""
    extra_flags = tuple(extra_flags)
    if not force and not _stale(extra_flags):
        return LIBPATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIBPATH):
            return LIBPATH
        raise RuntimeError("hipcc not found and no prebuilt libdaqp_amd.so")
    os.makedirs(OBJDIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(extra_flags):      # another process built it while this one waited
                return LIBPATH
            procs = []
            for unit in units(extra_flags):
                if force or _unit_stale(unit, extra_flags):
                    obj = os.path.join(OBJDIR, unit + ".o")
                    # the code generator's per-kernel register / scratch / LDS report is kept next to the object
                    # (kernel_resources(); tests/test_cpu.py holds the hot kernels to their budgets: an array that silently
                    # moves to scratch costs a factor, not a percent)
                    cmd = [hipcc, *HIPFLAGS, *extra_flags, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, unit), "-o", obj + ".tmp"]
                    if verbose:
                        print(" ".join(cmd))
                    log = open(obj + ".log.tmp", "w")
                    procs.append((cmd, obj, subprocess.Popen(cmd, stderr=log), log))
            for cmd, obj, pr, log in procs:
                rc = pr.wait()
                log.close()
                if rc != 0:
                    with open(obj + ".log.tmp") as fh:
                        print("".join(l for l in fh if "remark:" not in l and "[-Rpass-analysis" not in l)[-8000:])
                    raise subprocess.CalledProcessError(rc, cmd)
            for cmd, obj, pr, log in procs:
                os.replace(obj + ".tmp", obj)
                os.replace(obj + ".log.tmp", obj + ".resources.txt")
                with open(obj + ".flags", "w") as fh:
                    fh.write(_flag_key(extra_flags))
            tmp = LIBPATH + ".tmp.%d" % os.getpid()
            cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *[os.path.join(OBJDIR, u + ".o") for u in units(extra_flags)], "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIBPATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIBPATH


def kernel_resources():
    """{demangled-ish kernel name: dict(vgprs, agprs, sgprs, scratch, occupancy, lds, sgpr_spill, vgpr_spill)} of the last build,
    from the code generator's own report (-Rpass-analysis=kernel-resource-usage) kept per translation unit in lib/obj."""
    import re
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
            "LDS Size [bytes/block]": "lds"}
    out = {}
    for unit in units():
        path = os.path.join(OBJDIR, unit + ".o.resources.txt")
        if not os.path.exists(path):
            continue
        cur = None
        with open(path) as fh:
            for line in fh:
                m = re.search(r"remark:\s+Function Name: (\S+)", line)
                if m:
                    cur = out.setdefault(m.group(1), {})
                    continue
                m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+)", line)
                if m and cur is not None and m.group(1) in keys:
                    cur[keys[m.group(1)]] = int(m.group(2))
    try:
        names = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True, check=True).stdout.split("\n")
        out = {n.replace("daqp_amd::", "").split("(")[0].replace("void ", ""): v for n, v in zip(names, out.values())}
    except (OSError, subprocess.CalledProcessError):
        pass
    return out


_lib = None


def lib():
    """The loaded C ABI.  Raises if the HIP library is missing: there is nothing to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        build()
    # DAQP_AMD_LIBRARY: another build of the library (tools/variants.sh links tuning variants side by side)
    L = C.CDLL(os.environ.get("DAQP_AMD_LIBRARY") or LIBPATH)
    vp, ci = C.c_void_p, C.c_int
    L.daqp_amd_last_error.restype = C.c_char_p
    L.daqp_amd_version.restype = C.c_char_p
    L.daqp_batch_create.argtypes = [C.POINTER(vp), ci, ci, ci, ci, ci, C.POINTER(DAQPSettings), ci]
    L.daqp_batch_free.argtypes = [vp]
    L.daqp_batch_free.restype = None
    L.daqp_amd_release_pool.argtypes = []
    L.daqp_amd_release_pool.restype = None
    L.daqp_batch_set_stream.argtypes = [vp, vp]
    L.daqp_batch_set_stream.restype = None
    L.daqp_batch_set_settings.argtypes = [vp, C.POINTER(DAQPSettings)]
    L.daqp_batch_set_settings.restype = None
    L.daqp_batch_set_exact.argtypes = [vp, ci]
    L.daqp_batch_set_exact.restype = None
    L.daqp_batch_setup.argtypes = [vp, C.POINTER(DAQPBatchProblem), ci]
    L.daqp_batch_setup_shared.argtypes = [vp, C.POINTER(DAQPBatchProblem), ci]
    L.daqp_batch_update.argtypes = [vp, ci, C.POINTER(DAQPBatchProblem)]
    L.daqp_batch_solve.argtypes = [vp, C.POINTER(DAQPBatchResult)]
    L.daqp_batch_setup_flags.argtypes = [vp, c_int_p]
    L.daqp_batch_working_sets.argtypes = [vp, c_int_p, c_int_p]
    L.daqp_batch_set_primal_start.argtypes = [vp, C.c_void_p, ci]
    L.daqp_batch_prox_info.argtypes = [vp, c_int_p, c_int_p, C.c_void_p]
    L.daqp_quadprog_batch.argtypes = [C.POINTER(DAQPBatchResult), C.POINTER(DAQPBatchProblem), C.POINTER(DAQPSettings)]
    L.daqp_quadprog_batch_multi.argtypes = [C.POINTER(DAQPBatchResult), C.POINTER(DAQPBatchProblem), C.POINTER(DAQPSettings), c_int_p, ci]
    L.daqp_batch_create_multi.argtypes = [C.POINTER(vp), ci, ci, ci, ci, ci, C.POINTER(DAQPSettings), c_int_p, ci]
    L.daqp_batch_free_multi.argtypes = [vp]
    L.daqp_batch_free_multi.restype = None
    L.daqp_batch_multi_shards.argtypes = [vp]
    L.daqp_batch_multi_shard.argtypes = [vp, ci, c_int_p, c_int_p]
    L.daqp_batch_multi_shard.restype = vp
    L.daqp_batch_setup_multi.argtypes = [vp, C.POINTER(DAQPBatchProblem), ci]
    L.daqp_batch_update_multi.argtypes = [vp, ci, C.POINTER(DAQPBatchProblem)]
    L.daqp_batch_solve_multi.argtypes = [vp, C.POINTER(DAQPBatchResult)]
    L.daqp_batch_setup_multi_shards.argtypes = [vp, C.POINTER(DAQPBatchProblem), ci]
    L.daqp_batch_update_multi_shards.argtypes = [vp, ci, C.POINTER(DAQPBatchProblem)]
    L.daqp_batch_solve_multi_shards.argtypes = [vp, C.POINTER(DAQPBatchResult)]
    L.daqp_batch_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.daqp_batch_rechecked.argtypes = [vp]
    L.daqp_batch_set_recheck.argtypes = [vp, ci]
    L.daqp_batch_set_recheck.restype = None
    L.daqp_batch_recheck_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.daqp_batch_device_bytes.argtypes = [vp]
    L.daqp_batch_device_bytes.restype = C.c_ulonglong
    L.daqp_batch_enable_trace.argtypes = [vp, ci]
    L.daqp_batch_read_trace.argtypes = [vp, c_int_p]
    L.daqp_batch_read_ldp.argtypes = [vp, ci] + [c_double_p] * 6
    L.daqp_batch_enable_profile.argtypes = [vp, ci]
    L.daqp_batch_read_profile.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.daqp_quadprog.argtypes = [C.POINTER(DAQPResult), C.POINTER(DAQPProblem), C.POINTER(DAQPSettings)]
    L.daqp_quadprog.restype = None
    L.daqp_solve.argtypes = [C.POINTER(DAQPResult), vp]
    L.daqp_solve.restype = None
    L.setup_daqp.argtypes = [C.POINTER(DAQPProblem), vp, c_double_p]
    L.setup_daqp_main.argtypes = [C.POINTER(DAQPProblem), vp, c_double_p, ci]
    L.daqp_update_ldp.argtypes = [ci, vp, C.POINTER(DAQPProblem)]
    L.daqp_default_settings.argtypes = [C.POINTER(DAQPSettings)]
    L.daqp_default_settings.restype = None
    L.free_daqp_workspace.argtypes = [vp]
    L.free_daqp_workspace.restype = None
    L.free_daqp_ldp.argtypes = [vp]
    L.free_daqp_ldp.restype = None
    L.daqp_primal_init_active.argtypes = [C.POINTER(DAQPProblem), c_double_p]
    L.daqp_primal_init_active.restype = None
    L.daqp_dual_init_active.argtypes = [C.POINTER(DAQPProblem), c_double_p]
    L.daqp_dual_init_active.restype = None
    L.setup_daqp_ldp.argtypes = [vp, C.POINTER(DAQPProblem), ci]
    L.daqp_ldp.argtypes = [vp]
    L.ldp2qp_solution.argtypes = [vp]
    L.ldp2qp_solution.restype = None
    L.daqp_extract_result.argtypes = [C.POINTER(DAQPResult), vp]
    L.daqp_extract_result.restype = None
    L.daqp_minrep.restype = None
    _lib = L
    return L


def last_error():
    return lib().daqp_amd_last_error().decode()


def default_settings(**kw):
    s = DAQPSettings()
    lib().daqp_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise TypeError(f"unknown DAQP setting {k!r}")
        setattr(s, k, v)
    return s
