"""Where the wavefront-per-instance kernel stops paying at T = 30: warm launch times of both long-horizon
kernels over the batch size, each forced through the test-hook library:
MPCX_LIB=.../libmpcx_hooks.so MPCX_STAGEWISE=0|1 python scripts/wave_crossover.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine
from tests.helpers import make_problem
for name in ("dsr_ackermann", "dsr_diff"):
    for B in (1024, 2048, 3072, 4096, 6144, 8192, 12288):
        v, ov, x0, xref = make_problem(name, B, seed=3, T=30)
        x0, xref = engine._dev(x0), engine._dev(xref)
        cold = engine.solve_batch(v, x0, xref)
        up = cold["u"].clone()
        out = engine._outputs(B, v.nu, v.T, x0.device)
        for _ in range(10):
            engine.solve_batch(v, x0, xref, up, None, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            engine.solve_batch(v, x0, xref, up, None, out=out)
        e1.record(); torch.cuda.synchronize()
        print(f"{name} B={B}: {e0.elapsed_time(e1) / 40 * 1e3:.0f} us, statuses {np.bincount(out['status'].cpu().numpy(), minlength=5).tolist()}", flush=True)
