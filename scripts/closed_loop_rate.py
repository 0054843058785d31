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
