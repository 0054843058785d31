"""Phase timing of the wavefront-per-instance kernel from a -DMPCX_STAMPS build (MPCX_LIB points at it).
python scripts/stamps_wave.py [variant] [T] [B] [warm]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine, _lib
from tests.helpers import make_problem
name = sys.argv[1] if len(sys.argv) > 1 else "dsr_ackermann"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024   # what the chip holds at once at T = 30
v, ov, x0, xref = make_problem(name, B, T=T)
warm = len(sys.argv) > 4 and sys.argv[4] == "warm"
out = engine.solve_batch(v, x0, xref)
up = out["u"].cpu().numpy() if warm else None
for _ in range(2):
    out = engine.solve_batch(v, x0, xref, up)
torch.cuda.synchronize()
lib = _lib.load()
n = min(4096, B)
buf = (C.c_ulonglong * (n * 16))()
lib.mpcx_debug_stamps.restype = C.c_int
lib.mpcx_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mpcx_debug_stamps(C.cast(buf, C.c_void_p), n) == n
st = np.frombuffer(buf, dtype=np.uint64).reshape(n, 16).astype(np.int64)
print("status", np.bincount(out["status"].cpu().numpy(), minlength=5), "iters mean", out["iters"].float().mean().item())
ph = {"dynamics": st[:, 1] - st[:, 0], "condense": st[:, 2] - st[:, 1], "ipm0": st[:, 3] - st[:, 2], "pdas0": st[:, 4] - st[:, 3],
      "fallback": st[:, 5] - st[:, 4], "epilogue": st[:, 6] - st[:, 5], "total": st[:, 6] - st[:, 0],
      "last chol": st[:, 9] - st[:, 8], "last solve": st[:, 10] - st[:, 9],
      # the last full interior-point iteration: (11) matvec (12) residuals (13) factor (14) solve (15)
      "ipm matvec": st[:, 12] - st[:, 11], "ipm resid": st[:, 13] - st[:, 12], "ipm chol": st[:, 14] - st[:, 13], "ipm solve": st[:, 15] - st[:, 14]}
print("active-set rounds per instance: mean", st[:, 7].mean(), "hist", np.bincount(st[:, 7].astype(int)))
for nm, d in ph.items():
    print(f"{nm:10s} median {np.median(d):9.0f} mean {d.mean():9.0f} p99 {np.percentile(d, 99):9.0f} max {d.max():9.0f} cycles ")
it = out["iters"].cpu().numpy()
print("iterations: percentiles 50/90/99/100", np.percentile(it, [50, 90, 99, 100]).tolist())
