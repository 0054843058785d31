"""Back-to-back solves launched one by one against the same solves recorded once as a HIP graph:
what the launch path costs per step.  usage: graph_gap.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from model_predictive_control_amd import engine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
v, x0, xref, up, hp = bench.build_inputs(0, 65536)
out = engine.solve_batch(v, x0, xref, up, hp)
torch.cuda.synchronize()
def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); a.record(); fn(); b.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6, a.elapsed_time(b) / K * 1e3
def eager():
    for _ in range(K):
        engine.solve_batch(v, x0, xref, up, hp, out=out)
for _ in range(3):
    print("eager  us/step (wall, events): %.2f %.2f" % timed(eager))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    eager()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eager()
for _ in range(3):
    print("graph  us/step (wall, events): %.2f %.2f" % timed(g.replay))
