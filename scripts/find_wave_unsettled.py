"""Searches random controllers (the generator of tests/test_gpu_fuzz.py, horizons the
wavefront-per-instance kernel takes) for instances that kernel leaves INACCURATE, with the kernel
forced at every batch size through the test-hook library:
MPCX_LIB=.../libmpcx_hooks.so MPCX_STAGEWISE=0 python scripts/find_wave_unsettled.py [n_seeds] [B] [T for every seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine
from tests.helpers import make_problem
from tests.test_gpu_fuzz import BASES, _random_overrides
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
T_ALL = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bases = [b for b in BASES if b != "sim_ackermann"]
total = 0
SEEDS = [int(x) for x in os.environ["SEEDS"].split(",")] if os.environ.get("SEEDS") else range(n_seeds)
for seed in SEEDS:
    rng = np.random.default_rng(31000 + seed)
    base = bases[seed % len(bases)]
    kw = _random_overrides(rng, base)
    kw["T"] = int(rng.choice([3, 7, 8, 13, 18, 24, 30]))
    if T_ALL:
        kw["T"] = T_ALL
    v, ov, x0, xref = make_problem(base, B, seed=500 + seed, **kw)
    out = engine.solve_batch(v, x0, xref)
    st = out["status"].cpu().numpy()
    up = np.where(np.isnan(out["u"].cpu().numpy()), 0.0, out["u"].cpu().numpy())
    out2 = engine.solve_batch(v, x0, xref, up)
    st2 = out2["status"].cpu().numpy()
    total += 2 * B
    if (st == 1).any() or (st2 == 1).any():
        print("seed", seed, base, "T", kw["T"], "cold", np.flatnonzero(st == 1).tolist()[:8], "warm", np.flatnonzero(st2 == 1).tolist()[:8],
              "statuses", np.bincount(st, minlength=5).tolist(), np.bincount(st2, minlength=5).tolist(), flush=True)
    if (st >= 3).any() or (st2 >= 3).any():
        print("seed", seed, base, "T", kw["T"], "FAILED/nonfinite statuses", np.bincount(st, minlength=5).tolist(), np.bincount(st2, minlength=5).tolist(), flush=True)
print("solves", total)
