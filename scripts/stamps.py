"""Phase timing of the pair kernel from a -DMPCX_STAMPS build (MPCX_LIB points at it)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine, variants, workload, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "dsr_ackermann"
B = 65536
import bench
bench.VARIANT = name
v, x0, xref, up, hp = bench.build_inputs(0, B)     # tick 10 of a closed-loop run, as the benchmark times
for _ in range(3):
    out = engine.solve_batch(v, x0, xref, up, hp)
torch.cuda.synchronize()
lib = _lib.load()
n = 2048
buf = (C.c_ulonglong * (n * 16))()
lib.mpcx_debug_stamps.restype = C.c_int
lib.mpcx_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mpcx_debug_stamps(C.cast(buf, C.c_void_p), n) == n
st = np.frombuffer(buf, dtype=np.uint64).reshape(n, 16).astype(np.int64)
its = out["iters"].cpu().numpy().reshape(-1, 32).max(1)[:n]
print("wave iters (max over 32 instances): mean", its.mean(), "instance mean", out["iters"].float().mean().item(),
      "status counts", np.bincount(out["status"].cpu().numpy(), minlength=5))
fb = st[:, 5] > 0
end_solve = np.where(fb, st[:, 5], st[:, 3])
ph = {"load": st[:, 8] - st[:, 0], "checks": st[:, 9] - st[:, 8], "prepare_lin": st[:, 11] - st[:, 9], "stage_eval": st[:, 10] - st[:, 11], "condense": st[:, 1] - st[:, 10], "ipm0": st[:, 2] - st[:, 1], "pdas0": st[:, 3] - st[:, 2],
      "fallback": np.where(fb, st[:, 5] - st[:, 3], 0), "epilogue_compute": st[:, 12] - end_solve, "epilogue_store": st[:, 6] - st[:, 12], "total": st[:, 6] - st[:, 0]}
print("active-set rounds per wave: mean", st[:, 7].mean(), "hist", np.bincount(st[:, 7].astype(int)))
print("waves taking the fallback:", int(fb.sum()), "of", n)
for nm, d in ph.items():
    print(f"{nm:9s} median {np.median(d):9.0f} mean {d.mean():9.0f} cycles")
tot = (st[:, 6] - st[:, 0]).astype(float)
print("wavefront time, cycles: percentiles 1/50/90/99/100:", [int(np.percentile(tot, q)) for q in (1, 50, 90, 99, 100)],
      "(a launch of 2,048 wavefronts on 1,024 SIMDs takes two of them back to back)")
