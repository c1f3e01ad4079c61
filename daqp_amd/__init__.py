"""daqp_amd -- MI355X-native batched dual active-set QP path behind the DAQP C API.

The numerical work is done by hand-written HIP kernels in daqp_amd/csrc (one wavefront per QP),
reached only through the C ABI in include/daqp_amd.h.  This package is the host-side mirror of
the reference's Python binding for that path.
"""
from ._lib import build, default_settings, last_error, lib, LIBPATH  # noqa: F401
from .api import (BatchModel, MultiBatchModel, Model, solve, solve_batch, solve_batch_multi, UPDATE_Rinv, UPDATE_M, UPDATE_v, UPDATE_d,  # noqa: F401
                  UPDATE_sense, UPDATE_unconstrained, UPDATE_eliminate)

__all__ = ["build", "lib", "solve", "Model", "solve_batch", "solve_batch_multi", "BatchModel", "MultiBatchModel", "default_settings", "last_error"]
