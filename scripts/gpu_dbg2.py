"""Dump the instances where GPU and oracle disagree most (or the GPU reports failure)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine, variants, workload
from oracle import mpc_oracle as O, orc
name = sys.argv[1]; B = 65536
v = variants.get(name); ov = O.variant(name)
courses = workload.make_courses(32)
inst = workload.make_instances(courses, B, seed=7)
xref = workload.reference_window(courses, inst, v.T)
x0 = inst["state"][:, :3].copy()
ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=16)
out = engine.solve_batch(v, x0, xref)
up = ref["u"]; hp = (np.arange(B) % 3 != 0).astype(np.uint8)
ref2 = orc.iterative_batch(ov, x0, xref, up, hp, nthreads=16)
out2 = engine.solve_batch(v, x0, xref, up, hp)
res = {}
for tag, o, r in (("cold", out, ref), ("warm", out2, ref2)):
    st = o["status"].cpu().numpy(); u = o["u"].cpu().numpy()
    e = np.abs(u - r["u"]).reshape(B, -1).max(1); e[np.isnan(e)] = 0
    worst = np.argsort(-e)[:5]
    bad = np.nonzero(st > 1)[0][:5]
    print(tag, "worst", worst, e[worst], "gpu status", st[worst], "oracle status", r["status"][worst], "failed", bad, "oracle there", r["status"][bad])
    idx = np.unique(np.concatenate([worst, bad]))
    res[tag + "_idx"] = idx; res[tag + "_gpu_u"] = u[idx]; res[tag + "_ref_u"] = r["u"][idx]
    res[tag + "_gpu_st"] = st[idx]; res[tag + "_ref_st"] = r["status"][idx]; res[tag + "_iters"] = o["iters"].cpu().numpy()[idx]
res["x0"] = x0; res["xref"] = xref; res["up"] = up; res["hp"] = hp
os.makedirs("gpurun_out", exist_ok=True)
sel = np.unique(np.concatenate([res["cold_idx"], res["warm_idx"]]))
np.savez("gpurun_out/dbg2_%s.npz" % name, sel=sel, x0=x0[sel], xref=xref[sel], up=up[sel], hp=hp[sel], **{k: v_ for k, v_ in res.items() if k not in ("x0", "xref", "up", "hp")})
