"""Host-buffer round trip of the headline batch: H2D of the inputs, solve, D2H of the outputs
(pinned host memory), against the device-resident rate."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from model_predictive_control_amd import engine
B = 65536
v, x0, xref, up, hp = bench.build_inputs(0, B)
host = [t.cpu().pin_memory() for t in (x0, xref, up, hp)]
dev = [torch.empty_like(t) for t in (x0, xref, up, hp)]
out = engine.solve_batch(v, x0, xref, up, hp)
hout = {k: torch.empty_like(t, device="cpu").pin_memory() for k, t in out.items()}
def once():
    for d, h in zip(dev, host):
        d.copy_(h, non_blocking=True)
    engine.solve_batch(v, dev[0], dev[1], dev[2], dev[3], out=out)
    for k in out:
        hout[k].copy_(out[k], non_blocking=True)
for _ in range(3):
    once()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    once()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
nbytes = sum(t.numel() * t.element_size() for t in host) + sum(t.numel() * t.element_size() for t in hout.values())
print(f"host round trip: {dt*1e3:.3f} ms/batch = {B/dt:.3e} solves/s, {nbytes/1e6:.1f} MB moved = {nbytes/dt/1e9:.1f} GB/s over PCIe")
