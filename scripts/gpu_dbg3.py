"""Find instances whose outputs are NaN although the status says solved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine, variants, workload
name = sys.argv[1]; B = 65536
v = variants.get(name)
courses = workload.make_courses(32)
inst = workload.make_instances(courses, B, seed=7)
xref = workload.reference_window(courses, inst, v.T)
x0 = inst["state"][:, :3].copy()
out = engine.solve_batch(v, x0, xref)
st = out["status"].cpu().numpy(); u = out["u"].cpu().numpy(); x = out["x"].cpu().numpy(); it = out["iters"].cpu().numpy(); ob = out["obj"].cpu().numpy()
nan_u = np.isnan(u).reshape(B, -1).any(1); nan_x = np.isnan(x).reshape(B, -1).any(1)
print("status", np.bincount(st, minlength=5), "nan u", nan_u.sum(), "nan x", nan_x.sum(), "nan obj", np.isnan(ob).sum())
idx = np.nonzero(nan_u | nan_x)[0][:10]
for i in idx:
    print(i, "lane pair", i % 32, "wave", i // 32, "status", st[i], "iters", it[i], "u", u[i].round(4).tolist(), "x", x[i].round(3).tolist())
