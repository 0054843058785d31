"""Latency of one call of the reference-named functions (one robot), as a user of the reference
would make it: python scripts/single_call_latency.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import mpc_diff_drive as m, mpc_dsr, engine, variants
cx, cy, cyaw = m.makeEightShapeTrajectory(400, 10.0) if hasattr(m, "makeEightShapeTrajectory") else (None, None, None)
cyaw = m.smooth_yaw(cyaw)
state = m.State(x=cx[0], y=cy[0], yaw=cyaw[0], vx=0.0)
xref, tind = m.calc_ref_trajectory2(state, cx, cy, cyaw, 3, 0)
x0 = [state.x, state.y, state.yaw]
ov = ow = None
for _ in range(5):
    ov, ow, ox, oy, oyaw = m.iterative_linear_mpc_control(xref, x0, ov, ow)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    ov, ow, ox, oy, oyaw = m.iterative_linear_mpc_control(xref, x0, ov, ow)
dt = (time.perf_counter() - t0) / n
print(f"mpc_diff_drive.iterative_linear_mpc_control: {dt*1e6:.0f} us per call")
t0 = time.perf_counter()
for _ in range(n):
    xref, tind = m.calc_ref_trajectory2(state, cx, cy, cyaw, 3, 0)
dt = (time.perf_counter() - t0) / n
print(f"mpc_diff_drive.calc_ref_trajectory2: {dt*1e6:.0f} us per call")
v = variants.get("diff_drive")
x0d = torch.tensor([x0], dtype=torch.float64, device="cuda"); xrd = torch.as_tensor(np.asarray(xref)[None], dtype=torch.float64).cuda()
out = engine.solve_batch(v, x0d, xrd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    engine.solve_batch(v, x0d, xrd, out=out)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"engine.solve_batch B=1 resident, synchronised: {dt*1e6:.0f} us per call")
