"""What the other solvers say about the instances the wavefront-per-instance kernel reports FAILED
(seeds of scripts/find_wave_unsettled.py at T = 30; run with the test-hook library):
MPCX_LIB=.../libmpcx_hooks.so MPCX_STAGEWISE=0|1 python scripts/wave_failed_probe.py 18,78,108,233"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine
from oracle import orc
from tests.helpers import make_problem
from tests.test_gpu_fuzz import BASES, _random_overrides
bases = [b for b in BASES if b != "sim_ackermann"]
for seed in [int(x) for x in sys.argv[1].split(",")]:
    rng = np.random.default_rng(31000 + seed)
    base = bases[seed % len(bases)]
    kw = _random_overrides(rng, base)
    rng.choice([3, 7, 8, 13, 18, 24, 30])
    kw["T"] = 30
    v, ov, x0, xref = make_problem(base, 8192, seed=500 + seed, **kw)
    out = engine.solve_batch(v, x0, xref)
    up = np.where(np.isnan(out["u"].cpu().numpy()), 0.0, out["u"].cpu().numpy())
    out2 = engine.solve_batch(v, x0, xref, up)
    st2 = out2["status"].cpu().numpy()
    bad = np.flatnonzero(st2 != 0)
    ref = orc.iterative_batch(ov, x0[bad], xref[bad], up[bad], None, nthreads=8)
    du = np.abs(out2["u"].cpu().numpy()[bad] - ref["u"]).reshape(len(bad), -1).max(1)
    print("seed", seed, "instances", bad.tolist(), "engine", st2[bad].tolist(), "iters", out2["iters"].cpu().numpy()[bad].tolist(),
          "oracle", ref["status"].tolist(), "oracle iters", ref["iters"].tolist(), "max |du|", du.tolist(), flush=True)
