"""Random search windows against the oracle: courses of 1 to several hundred points, previous
indices from three in front of the row to three past its end, both reference-window flavours,
horizons 1 to 30, search widths 1 to 33.  Indices and windows must be identical, distances equal to
rounding.  python scripts/fuzz_nearest.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine, workload
from oracle import orc
bad = 0
for seed in range(40):
    rng = np.random.default_rng(900 + seed)
    base = workload.make_courses(4, seed=seed)
    courses = []
    for cx, cy, cyaw in base:
        n = int(rng.choice([1, 2, 5, 40, len(cx)]))
        courses.append((np.asarray(cx[:n]), np.asarray(cy[:n]), np.asarray(cyaw[:n])))
    tab = engine.CourseTable(courses)
    cxa, cya, cwa = tab.cx.cpu().numpy(), tab.cy.cpu().numpy(), tab.cyaw.cpu().numpy()
    B = 257
    cid = rng.integers(0, 4, B).astype(np.int32)
    lens = tab.clen_host[cid]
    pind = rng.integers(-3, lens + 3).astype(np.int32)
    st = np.stack([rng.uniform(-5, 30, B), rng.uniform(-8, 8, B), rng.uniform(-4, 4, B), rng.uniform(0, 2.0, B)], axis=1)
    for mode, param in ((0, float(rng.choice([0.1, 0.5, 1.0]))), (1, float(rng.integers(1, 5)))):
        T = int(rng.choice([1, 5, 10, 30])); ns = int(rng.choice([1, 3, 10, 16, 33]))
        ind, mind, xr = engine.nearest_ref_batch(mode, T, 0.2, param, st, cid, pind, tab, n_search=ns)
        wi, wm, wx = orc.nearest_ref_batch(mode, T, 0.2, param, st, cid, pind, cxa, cya, cwa, tab.clen_host, n_search=ns)
        gi, gx, gm = ind.cpu().numpy(), xr.cpu().numpy(), mind.cpu().numpy()
        ok = np.array_equal(gi, wi) and np.array_equal(np.nan_to_num(gx, nan=-7.0), np.nan_to_num(wx, nan=-7.0))
        okm = np.allclose(np.nan_to_num(gm, nan=-7.0), np.nan_to_num(wm, nan=-7.0), rtol=1e-12, atol=1e-12)
        if not (ok and okm):
            bad += 1
            k = np.nonzero(gi != wi)[0]
            print("FLAG seed", seed, "mode", mode, "T", T, "ns", ns, "param", param, "ind mismatch at", k[:5], gi[k[:5]], wi[k[:5]], "pind", pind[k[:5]], "len", lens[k[:5]], "xref equal", np.array_equal(np.nan_to_num(gx, nan=-7.0), np.nan_to_num(wx, nan=-7.0)), "mind ok", okm)
print("done, flagged", bad)
