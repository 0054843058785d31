"""ctypes binding of libmpcx.so (the HIP engine).  There is no CPU fallback: if the
library is missing or no GPU is present, calls raise."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPCX_LIB selects another build of the same library (kernel experiments); never a fallback
LIB_PATH = os.environ.get("MPCX_LIB") or os.path.join(_HERE, "lib", "libmpcx.so")

ABI_VERSION = 2

ST_OK, ST_INACCURATE, ST_INFEASIBLE, ST_NONFINITE, ST_FAILED = 0, 1, 2, 3, 4
MODEL_DIFF_DRIVE, MODEL_DSR = 0, 1
EQ_NONE, EQ_ACKERMANN, EQ_CRAB, EQ_FIXED_STEER = 0, 1, 2, 3
MAX_T = 64
E_ARG, E_UNSUPPORTED, E_HIP, E_NOMEM = -1, -2, -3, -4


class MpcxError(RuntimeError):
    pass


class CVariant(C.Structure):
    """Mirror of `mpcx_variant` in include/mpcx.h."""
    _fields_ = [
        ("model", C.c_int32), ("nu", C.c_int32), ("T", C.c_int32), ("eq_mode", C.c_int32),
        ("chain", C.c_int32), ("pad_", C.c_int32),
        ("dt", C.c_double), ("wb", C.c_double),
        ("R", C.c_double * 4), ("Rd", C.c_double * 4),
        ("Q", C.c_double * 3), ("q_growth", C.c_double), ("Qf", C.c_double * 3),
        ("umax", C.c_double * 4), ("dumax", C.c_double * 4), ("first_step", C.c_double * 4),
        ("steer_fixed", C.c_double),
        ("cold_start", C.c_double * 4),
    ]


_PD = C.c_void_p   # device pointers travel as plain addresses

# name -> (restype, argtypes); every symbol include/mpcx.h declares
SIGNATURES = {
    "mpcx_abi_version": (C.c_int, []),
    "mpcx_last_error": (C.c_char_p, []),
    "mpcx_variant_by_name": (C.c_int, [C.c_char_p, C.POINTER(CVariant)]),
    "mpcx_workspace_bytes": (C.c_int64, [C.POINTER(CVariant), C.c_int64]),
    "mpcx_solve_batch": (C.c_int, [C.POINTER(CVariant), C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD]),
    "mpcx_solve_batch_noalloc": (C.c_int, [C.POINTER(CVariant), C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD]),
    "mpcx_static_workspace_bytes": (C.c_int64, []),
    "mpcx_solve_range_batch": (C.c_int, [C.POINTER(CVariant), C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD]),
    "mpcx_solve_modes_batch": (C.c_int, [_PD, C.c_int32, _PD, _PD, C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD]),
    "mpcx_mode_sort_scratch_ints": (C.c_int64, [C.c_int64, C.c_int32]),
    "mpcx_mode_sort_batch": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, _PD, _PD, _PD, _PD, C.c_int64, _PD, _PD, _PD, _PD,
                                       _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD]),
    "mpcx_mode_unsort_batch": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, _PD, _PD, _PD, _PD, _PD, _PD]),
    "mpcx_segment_step_batch": (C.c_int, [C.c_int64, C.c_int32, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD,
                                          C.c_int64, C.c_int64, _PD, _PD, _PD, _PD]),
    "mpcx_linear_mpc_batch": (C.c_int, [C.POINTER(CVariant), C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD]),
    "mpcx_rollout_batch": (C.c_int, [C.POINTER(CVariant), C.c_int64, _PD, _PD, _PD, _PD]),
    "mpcx_linearize_batch": (C.c_int, [C.POINTER(CVariant), C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD]),
    "mpcx_nearest_ref_batch": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                         C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD, _PD, C.c_int64, C.c_int64,
                                         _PD, _PD, _PD, _PD, _PD]),
    "mpcx_build_courses_batch": (C.c_int, [C.c_int64, C.c_int32, _PD, _PD, _PD, C.c_double, C.c_int64, C.c_int32,
                                           _PD, _PD, _PD, _PD, _PD, _PD]),
    "mpcx_advance_batch": (C.c_int, [C.POINTER(CVariant), C.c_int64, C.c_double, C.c_double, _PD, _PD, _PD, _PD,
                                     _PD, _PD, _PD, C.c_int64, C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD, _PD]),
}

_lib = None


def load():
    """Loads libmpcx.so once and types its entry points."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (first, so libmpcx.so binds to the HIP runtime torch ships, not a second one)
    if not os.path.exists(LIB_PATH):
        raise MpcxError(
            f"{LIB_PATH} is missing: build it with `python -m model_predictive_control_amd.build` "
            "(needs hipcc); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mpcx_abi_version() != ABI_VERSION:
        raise MpcxError("libmpcx.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().mpcx_last_error()
        raise MpcxError(f"mpcx error {rc}: {msg.decode() if msg else ''}")
