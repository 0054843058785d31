"""Host-buffer round trip of the headline batch against the device-resident rate: H2D of the
inputs, solve, D2H of the outputs through pinned memory, first one batch at a time on one
stream, then through engine.HostPipeline (three slots, a stream each)."""
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
print(f"one stream: {dt*1e3:.3f} ms/batch = {B/dt:.3e} solves/s, {nbytes/1e6:.1f} MB moved = {nbytes/dt/1e9:.1f} GB/s over PCIe")

pipe = engine.HostPipeline(v, B, depth=3)
ref_u = out["u"].cpu().numpy()
for s in pipe._slots:      # the caller's data, written into the pinned buffers once (a real caller fills them per batch)
    s.x0[:] = host[0].numpy(); s.xref[:] = host[1].numpy(); s.uprev[:] = host[2].numpy(); s.has_prev[:] = host[3].numpy()
for _ in range(6):
    pipe.submit(pipe.acquire())
torch.cuda.synchronize()
n = 60
t0 = time.perf_counter()
last = None
for _ in range(n):
    last = pipe.submit(pipe.acquire())
res = pipe.result(last)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
assert np.array_equal(res["u"], ref_u)
print(f"pipeline of 3: {dt*1e3:.3f} ms/batch = {B/dt:.3e} solves/s, {nbytes/dt/1e9:.1f} GB/s over PCIe (both directions together)")
