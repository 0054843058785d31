"""Latency of small batches at long horizons (the wavefront-per-instance kernel followed by the
stage-wise pass over what it left unsettled): warm solves, mean of 200 after 50.
python scripts/time_small_batches.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine
from tests.helpers import make_problem
for name, T, B in (("dsr_ackermann", 30, 1), ("dsr_ackermann", 30, 256), ("dsr_ackermann", 30, 1024), ("dsr_diff", 30, 1024),
                   ("dsr_ackermann", 12, 3072), ("diff_drive", 20, 512)):
    v, ov, x0, xref = make_problem(name, B, seed=3, T=T)
    x0, xref = engine._dev(x0), engine._dev(xref)
    cold = engine.solve_batch(v, x0, xref)
    up = cold["u"].clone()
    out = engine._outputs(B, v.nu, v.T, x0.device)
    for _ in range(50):
        engine.solve_batch(v, x0, xref, up, None, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        engine.solve_batch(v, x0, xref, up, None, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"{name} T={T} B={B}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per call, statuses {np.bincount(out['status'].cpu().numpy(), minlength=5).tolist()}")
