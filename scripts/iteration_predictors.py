"""Could instances be grouped so that the 64 lanes of a wavefront finish together?  Per-lane
interior-point iterations of the T = 30 batch against quantities known before the solve; for each,
the mean over wavefronts of the slowest lane if the batch were sorted by it (the ideal: sorted by
the iteration count itself).  Measured: cold 5.95 -> 5.19 by the distance to the end of the
reference window (ideal 4.88), warm 6.08 -> 6.03 by anything (ideal 5.12): not worth a gather.
    python scripts/iteration_predictors.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine, variants, workload
name, B, T = "dsr_ackermann", 65536, 30
v = variants.get(name, T=T)
courses = workload.make_courses(64, seed=1001)
inst = workload.make_instances(courses, B, seed=2003)
xref = workload.reference_window(courses, inst, v.T)
x0 = inst["state"][:, :3].copy()
out = engine.solve_batch(v, x0, xref)
up = out["u"].clone()
for label, u in (("cold", None), ("warm", up)):
    o = engine.solve_batch(v, x0, xref, u)
    it = o["iters"].cpu().numpy()
    def wavemax(order):
        return it[order].reshape(-1, 64).max(axis=1).mean()
    e0 = np.hypot(xref[:, 0, 0] - x0[:, 0], xref[:, 1, 0] - x0[:, 1])
    eT = np.hypot(xref[:, 0, -1] - x0[:, 0], xref[:, 1, -1] - x0[:, 1])
    dyaw = np.abs(((xref[:, 2, 0] - x0[:, 2]) + np.pi) % (2 * np.pi) - np.pi)
    usat = (np.abs(o["u"].cpu().numpy()) > 0.999 * np.asarray([1.0, 0.7853981633974483, 1.0, 0.7853981633974483])[None, :, None]).sum(axis=(1, 2)) if True else None
    print(label, "lane mean %.2f  wave-max mean as is %.2f  sorted by iters (ideal) %.2f" % (it.mean(), wavemax(np.arange(B)), wavemax(np.argsort(it))))
    for nm, p in (("pos err now", e0), ("pos err end", eT), ("yaw err", dyaw), ("saturated controls in answer", usat)):
        print("   sorted by", nm, "%.2f" % wavemax(np.argsort(p)), " corr %.2f" % np.corrcoef(p, it)[0, 1])
    if u is not None:
        it0 = out["iters"].cpu().numpy()
        print("   sorted by the cold solve's iterations %.2f" % wavemax(np.argsort(it0, kind="stable")), " corr %.2f" % np.corrcoef(it0, it)[0, 1])
