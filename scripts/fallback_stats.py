"""How often the lane-pair / four-lane kernels leave their first attempt (fixed interior-point
iterations + active-set rounds) unsettled, from a -DMPCX_DIAG build (MPCX_LIB points at it): the
iteration count then carries 100 x rounds of the first attempt, 10,000 x (rounds + 1) of each later one and
1,000,000 x (attempt + 1) per attempt left unsettled (lane-pair kernel: three attempts, 6,000,000 in all when
none settles; four-lane kernel: two, 3,000,000).
python scripts/fallback_stats.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from model_predictive_control_amd import engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for name in ("dsr_ackermann", "dsr_crab", "dsr_diff", "diff_drive", "mpc_py"):
    bench.VARIANT = name
    v, x0, xref, up, hp = bench.build_inputs(0, B)
    for tag, u, h in (("cold", None, None), ("warm", up, hp)):
        out = engine.solve_batch(v, x0, xref, u, h)
        it = out["iters"].cpu().numpy().astype(np.int64)
        st = out["status"].cpu().numpy()
        per = 32 if v.T == 5 else 16
        unsettled0 = (it // 1000000) >= 1
        unsettled1 = (it // 1000000) >= (6 if v.T == 5 else 3)
        second = (it // 1000000) >= 3
        rounds0 = (it % 10000) // 100
        n = (B // per) * per
        waves_fb = unsettled0[:n].reshape(-1, per).any(axis=1)
        print(f"{name:14s} {tag}: instances leaving attempt 0 unsettled {int(unsettled0.sum())} ({unsettled0.mean():.2e}), leaving the second too {int(second.sum())}, ending at the interior-point iterate {int(unsettled1.sum())}; "
              f"wavefronts taking the second attempt {int(waves_fb.sum())} of {waves_fb.size}; rounds in attempt 0: mean {rounds0.mean():.2f}, per-wave max mean "
              f"{rounds0[:n].reshape(-1, per).max(axis=1).mean():.2f}; status {np.bincount(st, minlength=5).tolist()}")
