"""The engine against the reference's own (x, u) programme (tests/xu_problem.py) on more instances
than the tests take: per variant, cold and warm, the (x, u) KKT certificate on every instance and scipy
SLSQP on the first n_scipy.  Prints one line per case.  python scripts/xu_parity_sweep.py [B] [n_scipy]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import xu_checks
from tests.test_gpu_xu_problem import CASES, EngineOps, _inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_scipy = int(sys.argv[2]) if len(sys.argv) > 2 else 128
for name, T in CASES + [("dsr_ackermann", 30), ("dsr_diff", 30)]:
    b = B if T < 30 else max(64, B // 8)
    ns = n_scipy if T < 30 else 0
    v, x0, xref = _inputs(name, T, b)
    t0 = time.time()
    wc, ww = xu_checks.run_variant(name, T, EngineOps(v), x0, xref, list(v.cold_start)[:v.nu], v.chain > 1, n_scipy=ns, controls=True)
    for tag, w in (("cold", wc), ("warm", ww)):
        print(f"{name:14s} T={T:2d} {tag}: {b} instances certified in (x, u): stationarity {w['stat']:.1e}, equalities {w['eq']:.1e}, "
              f"inequalities {w['viol']:.1e}, reported objective vs cost expression {w['dobj_out']:.1e};  scipy SLSQP on {w['n_scipy']}: "
              f"controls {w['du']:.1e}, objective {w['dobj']:.1e}, engine above scipy by {w['above']:.1e}", flush=True)
    print(f"   ({time.time() - t0:.0f} s)", flush=True)
