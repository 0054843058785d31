"""More seeds of tests/test_gpu_fuzz.py::test_random_controller_at_horizon_10 (the four-lane kernel),
failures printed instead of raised.  python scripts/fuzz_quad.py first_seed n_seeds"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine
from oracle import orc
from tests.helpers import compare, make_problem
from tests.test_gpu_fuzz import _horizon_10_case
first, n = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + n):
    base, kw, B = _horizon_10_case(seed)
    v, ov, x0, xref = make_problem(base, B, seed=90 + seed, **kw)
    try:
        ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=16)
        out = engine.solve_batch(v, x0, xref)
        d1 = compare(out, ref, u_tol=1e-6, x_tol=1e-6, obj_tol=1e-8, problem=dict(ov=ov, x0=x0, xref=xref))
        hp = (np.arange(B) % 3 != 0).astype(np.uint8)
        up = np.where(np.isnan(ref["u"]), 0.0, ref["u"])
        ref2 = orc.iterative_batch(ov, x0, xref, up, hp, nthreads=16)
        out2 = engine.solve_batch(v, x0, xref, up, hp)
        d2 = compare(out2, ref2, u_tol=1e-6, x_tol=1e-6, obj_tol=1e-8, problem=dict(ov=ov, x0=x0, xref=xref, uref=up, has_prev=hp))
        print("seed", seed, base, "B", B, "ok du %.1e / %.1e" % (d1[0], d2[0]), "status", np.bincount(out2["status"].cpu().numpy(), minlength=5).tolist(), flush=True)
    except AssertionError as e:
        bad += 1
        print("seed", seed, base, "B", B, "FLAG", str(e)[:300], flush=True)
print("flagged", bad, "of", n)
