"""Replays seeds of tests/test_gpu_fuzz.py with the comparison printed instead of asserted (statuses,
worst control difference, certificate of both answers).  python scripts/fuzz_replay.py seed..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine
from oracle import mpc_oracle as O, orc
from tests.helpers import make_problem, condense_batch, free_controls, kkt_certificate
from tests.test_gpu_fuzz import BASES, _random_overrides
for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(1000 + seed)
    base = BASES[seed % len(BASES)]
    kw = _random_overrides(rng, base)
    B = int(rng.choice([1, 7, 64, 65, 130, 1200, 3500]))
    B = int(os.environ.get("FUZZ_B", B))      # e.g. FUZZ_B=32768: rare lanes need large batches to show
    if "FUZZ_T" in os.environ:                 # e.g. FUZZ_T=30: the horizon the generator does not draw
        kw["T"] = int(os.environ["FUZZ_T"])
    v, ov, x0, xref = make_problem(base, B, seed=50 + seed, **kw)
    T = ov.T
    print("seed", seed, base, "T", T, "B", B, "chain", ov.chain, {k: kw[k] for k in ("dt", "q_growth") if k in kw})
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=16)
    out = engine.solve_batch(v, x0, xref)
    hp = (np.arange(B) % 3 != 0).astype(np.uint8)
    up = np.where(np.isnan(ref["u"]), 0.0, ref["u"])
    ref2 = orc.iterative_batch(ov, x0, xref, up, hp, nthreads=16)
    out2 = engine.solve_batch(v, x0, xref, up, hp)
    cold = np.tile(np.asarray(ov.cold_start, float)[:ov.nu, None], (1, T))
    for label, o, r, ur in (("cold", out, ref, np.tile(cold[None], (B, 1, 1))), ("warm", out2, ref2, np.where(hp[:, None, None] != 0, up, cold[None]))):
        u = o["u"].cpu().numpy(); st = o["status"].cpu().numpy()
        du = np.abs(u - r["u"]).reshape(B, -1).max(axis=1)
        dobj = np.abs(o["obj"].cpu().numpy() - r["obj"]) / np.maximum(1, np.abs(r["obj"]))
        print(" ", label, "status gpu", np.bincount(st, minlength=5).tolist(), "ref", np.bincount(r["status"], minlength=5).tolist(), "du max", np.nanmax(du), "dobj", np.nanmax(dobj), "iters max", o["iters"].cpu().numpy().max())
        if not np.array_equal(st, r["status"]) or np.nanmax(du) > 1e-6:
            print("FLAG seed", seed, label, "statuses differ" if not np.array_equal(st, r["status"]) else "", "du", np.nanmax(du))
        if ov.chain <= 1:
            ok = (st <= 1) & (r["status"] <= 1)
            xb = orc.rollout_batch(ov, x0[ok], ur[ok])
            H, g, lo, hi, rr, feas = condense_batch(ov, xref[ok], xb[:, 2, :T], x0[ok], ur[ok])
            for nm, uu in (("gpu", u), ("ref", r["u"])):
                w = free_controls(ov, uu[ok])
                stat, viol, neg = kkt_certificate(H, g, lo, hi, rr, w)
                f = 0.5 * np.einsum("bi,bij,bj->b", w, H, w) + np.einsum("bi,bi->b", g, w)
                k = np.nanargmax(du[ok])
                print("    ", nm, "cert stat %.2e viol %.2e neg %.2e" % (stat.max(), viol.max(), neg.max()), " at worst-du instance: stat %.2e f %.12e" % (stat[k], f[k]))
                if max(stat.max(), neg.max()) > 1e-8 or viol.max() > 1e-9:
                    print("FLAG seed", seed, label, nm, "certificate", stat.max(), viol.max(), neg.max())
