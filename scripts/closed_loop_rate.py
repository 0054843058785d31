"""Closed-loop throughput: vehicle-ticks per second with and without graph replay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine, variants, workload
name = sys.argv[1] if len(sys.argv) > 1 else "dsr_ackermann"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
v = variants.get(name)
courses = workload.make_courses(1024, seed=1001)
inst = workload.make_instances(courses, B, seed=2003)
table = engine.CourseTable(courses)
loop = engine.ClosedLoop(v, table, inst["state"][:, :4].copy(), inst["course_id"], pind0=inst["pind"])
loop.run(5)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter(); loop.run(n); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"{name} B={B}: eager  {dt*1e6:8.1f} us/tick  {B/dt:.3e} vehicle-ticks/s")
if hasattr(loop, "capture"):
    loop.capture()
    loop.run(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop.run(n); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name} B={B}: graph  {dt*1e6:8.1f} us/tick  {B/dt:.3e} vehicle-ticks/s")
print("done", int((loop.done != 0).sum()), "ticks", int(loop.ticks.max()))

# a fleet whose courses change motion mode (mpc_sim's three modes), device-resident segment hand-over
from model_predictive_control_amd import mpc_sim
from oracle import mpc_oracle as O
rows, seg_mode = [], []
for k, (cx, cy, cyaw) in enumerate(courses[:256]):
    n3 = len(cx) // 3
    for s_, (a, b_) in enumerate(((0, n3), (n3, 2 * n3), (2 * n3, len(cx)))):
        rows.append((cx[a:b_], cy[a:b_], np.array(O.smooth_yaw(list(cyaw[a:b_])))))
        seg_mode.append((k + s_) % 3)
table3 = engine.CourseTable(rows)
rng = np.random.default_rng(4)
cid = rng.integers(0, 256, B)
seg_start, seg_count = 3 * cid, np.full(B, 3)
tx, ty, tw = table3.cx.cpu().numpy(), table3.cy.cpu().numpy(), table3.cyaw.cpu().numpy()
state = np.stack([tx[seg_start, 0] + rng.normal(0, 0.1, B), ty[seg_start, 0] + rng.normal(0, 0.1, B), tw[seg_start, 0], np.zeros(B)], axis=1)
ctl = mpc_sim.MPC()
for resident in (True, False):
    sl = engine.SegmentedLoop(ctl.segment_variants(), table3, seg_mode, seg_start, seg_count, state, 40,
                              goal_dis=ctl.xy_tolerance_, stop_speed=ctl.stop_speed_)
    sl.device_resident = resident
    sl.run(30); torch.cuda.synchronize()      # segments of 40 ticks: the fleet spreads over all three modes
    if resident:
        torch.cuda.set_sync_debug_mode("error")
    t0 = time.perf_counter(); sl.run(n); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    torch.cuda.set_sync_debug_mode("default")
    modes = torch.bincount(sl.seg_mode[sl.course_id.long()], minlength=3).tolist()
    print(f"mixed-mode fleet (mpc_sim chain of 5 QPs per tick) B={B}, {'device-resident' if resident else 'host index lists'}: "
          f"{dt*1e6:8.1f} us/tick  {B/dt:.3e} vehicle-ticks/s; vehicles per mode {modes}, finished {int((sl.finished != 0).sum())}")
