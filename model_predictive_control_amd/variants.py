"""The reference's QP variants as `mpcx_variant` structs.  The table itself lives in the
C library (mpcx_variant_by_name); this module adds overrides, like the reference's
module globals allow (`module.T = 30` works there because T is read at call time)."""
from __future__ import annotations

import ctypes as C
from typing import Iterable

from . import _lib

NAMES = ("diff_drive", "dsr_ackermann", "dsr_crab", "dsr_diff",
         "sim_ackermann", "sim_diff", "sim_crab", "mpc_py")

_ARRAY_FIELDS = {"R": 4, "Rd": 4, "Q": 3, "Qf": 3, "umax": 4, "dumax": 4, "first_step": 4, "cold_start": 4}
_SCALAR_FIELDS = ("model", "nu", "T", "eq_mode", "chain", "dt", "wb", "q_growth", "steer_fixed")


def get(name: str, **overrides) -> _lib.CVariant:
    """Variant `name` with fields overridden (arrays take any sequence of the right length)."""
    v = _lib.CVariant()
    _lib.check(_lib.load().mpcx_variant_by_name(name.encode(), C.byref(v)))
    if "T" in overrides and v.chain > 1 and "chain" not in overrides:
        overrides["chain"] = overrides["T"]   # mpc_sim.py:295 chains over horizon_
    for key, val in overrides.items():
        if key in _ARRAY_FIELDS:
            vals = list(val) if isinstance(val, Iterable) else [val] * _ARRAY_FIELDS[key]
            if len(vals) > _ARRAY_FIELDS[key]:
                raise ValueError(f"{key}: too many values")
            arr = getattr(v, key)
            for i, x in enumerate(vals):
                arr[i] = float(x)
        elif key in _SCALAR_FIELDS:
            setattr(v, key, val)
        else:
            raise KeyError(key)
    return v


def copy(v: _lib.CVariant, **overrides) -> _lib.CVariant:
    out = _lib.CVariant()
    C.memmove(C.byref(out), C.byref(v), C.sizeof(v))
    for key, val in overrides.items():
        if key in _ARRAY_FIELDS:
            arr = getattr(out, key)
            for i, x in enumerate(val):
                arr[i] = float(x)
        else:
            setattr(out, key, val)
    return out
