"""Per-step time of the headline solve over a long run of back-to-back launches, in groups of 50:
how long the GPU takes to reach its sustained clock after idling.  usage: clock_ramp.py [groups]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from model_predictive_control_amd import engine
G = int(sys.argv[1]) if len(sys.argv) > 1 else 40
v, x0, xref, up, hp = bench.build_inputs(0, 65536)
out = engine.solve_batch(v, x0, xref, up, hp)
torch.cuda.synchronize()
time.sleep(2.0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(G + 1)]
ev[0].record()
for g in range(G):
    for _ in range(50):
        engine.solve_batch(v, x0, xref, up, hp, out=out)
    ev[g + 1].record()
torch.cuda.synchronize()
print("us/step per group of 50:", " ".join("%.1f" % (ev[g].elapsed_time(ev[g + 1]) / 50 * 1e3) for g in range(G)))
