"""Where a closed loop of chained (mpc_sim) solves and the oracle's loop part ways: runs both
tick by tick from the SAME state each tick (the oracle's), and for the vehicles whose one-tick
answers differ, re-enacts the chain link by link on both sides and against the
extended-precision minimiser of the reference's (x, u) programme.
usage: python scripts/chain_divergence.py [variant] [B] [ticks]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine, variants, workload  # noqa: E402
from oracle import mpc_oracle as O, orc  # noqa: E402
from tests import xu_checks, xu_problem as XU  # noqa: E402
from tests.test_gpu_xu_problem import EngineOps  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sim_crab"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 96
n = int(sys.argv[3]) if len(sys.argv) > 3 else 15
v, ov = variants.get(name), O.variant(name)
one = O.variant(name, chain=1)
courses = workload.make_courses(6, seed=31)
inst = workload.make_instances(courses, B, seed=32)
table = engine.CourseTable(courses)
tabs = (table.cx.cpu().numpy(), table.cy.cpu().numpy(), table.cyaw.cpu().numpy(), table.clen_host)
state = inst["state"][:, :4].copy()
want = O.closed_loop_batch(ov, *tabs, state, inst["course_id"], n, pind0=inst["pind"])
# replay the oracle's loop tick by tick, solving each tick's problem on the GPU as well
log = want["log"]
tind = inst["pind"].astype(np.int32)
st = state.copy()
for b in range(B):
    d = st[b, 2] - tabs[2][inst["course_id"][b], 0]
    st[b, 2] += -2 * np.pi if d >= np.pi else (2 * np.pi if d <= -np.pi else 0.0)
tind, _, _ = orc.nearest_ref_batch(1, ov.T, ov.dt, 2, st, inst["course_id"], tind, *tabs)
uprev = np.zeros((B, 4, ov.T)); has_prev = np.zeros(B, np.uint8)
ops = EngineOps(v)
T = ov.T
for k in range(n):
    st = log[k]
    tind, _, xref = orc.nearest_ref_batch(1, ov.T, ov.dt, 2, st, inst["course_id"], tind, *tabs)
    x0 = st[:, :3].copy()
    ref = orc.iterative_batch(ov, x0, xref, uprev, has_prev)
    gpu = {kk: t.cpu().numpy() for kk, t in engine.solve_batch(v, x0, xref, uprev, has_prev).items()}
    du = np.abs(gpu["u"] - ref["u"]).reshape(B, -1).max(axis=1)
    worst = int(np.nanargmax(du))
    print(f"tick {k}: one-tick |du| max {np.nanmax(du):.2e} (vehicle {worst}), status gpu {np.bincount(gpu['status'], minlength=5)} oracle {np.bincount(ref['status'], minlength=5)}")
    if np.nanmax(du) > 1e-8:
        b = worst
        prev = np.where(has_prev[b], uprev[b], 0.0)[None]
        xb = ops.rollout(x0[b:b + 1], prev)
        urG = np.zeros((1, 4, T)); urO = np.zeros((1, 4, T)); urA = np.zeros((1, 4, T))
        pG, pO, pA = prev.copy(), prev.copy(), prev.copy()
        for i in range(T):
            urG[:, :, i], urO[:, :, i], urA[:, :, i] = pG[:, :, i], pO[:, :, i], pA[:, :, i]
            lg = ops.linear(xref[b:b + 1], xb, x0[b:b + 1], urG)
            lo = orc.linear_mpc_batch(one, xref[b:b + 1], xb, x0[b:b + 1], urO)
            la = ops.linear(xref[b:b + 1], xb, x0[b:b + 1], urA)
            abc = xu_checks.stage_matrices(ops, xb, urA)
            p = XU.build(name, xref[b], x0[b], urA[0], abc[0], T=T)
            xs, us, kk = XU.arbiter(p, la["x"][0], la["u"][0])
            pg = XU.build(name, xref[b], x0[b], urG[0], xu_checks.stage_matrices(ops, xb, urG)[0], T=T)
            po = XU.build(name, xref[b], x0[b], urO[0], xu_checks.stage_matrices(ops, xb, urO)[0], T=T)
            print(f"   link {i}: gpu-arb {np.abs(lg['u'][0] - us).max():.2e} oracle-arb {np.abs(lo['u'][0] - us).max():.2e} "
                  f"arbiter kkt {kk[0]:.1e}/{kk[2]:.1e}; own-problem kkt: gpu {XU.kkt(pg, lg['x'][0], lg['u'][0])[0]:.1e} "
                  f"oracle {XU.kkt(po, lo['x'][0], lo['u'][0])[0]:.1e}; status gpu {lg['status'][0]} oracle {lo['status'][0]}; "
                  f"|uref gpu-oracle| {np.abs(urG - urO).max():.1e}")
            pG, pO, pA = lg["u"], lo["u"], us[None]
    ok = ref["status"] <= 1
    has_prev = ok.astype(np.uint8)
    uprev = ref["u"]
