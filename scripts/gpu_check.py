"""Quick GPU sanity run: parity of the HIP engine against the C oracle + rough timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine, variants, workload
from oracle import mpc_oracle as O, orc

def run(name, B, T=None, warm=True):
    kw = {} if T is None else {"T": T}
    v = variants.get(name, **kw)
    ov = O.variant(name, **kw)
    courses = workload.make_courses(32)
    inst = workload.make_instances(courses, B, seed=7)
    xref = workload.reference_window(courses, inst, v.T)
    x0 = inst["state"][:, :3].copy()
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
    out = engine.solve_batch(v, x0, xref)
    torch.cuda.synchronize()
    def cmp(tag, out, ref):
        st = out["status"].cpu().numpy(); u = out["u"].cpu().numpy(); x = out["x"].cpu().numpy(); ob = out["obj"].cpu().numpy()
        ok = (ref["status"] == 0) & (st <= 1)   # an oracle solve that calls itself inaccurate is no yardstick
        print(f"{name} T={v.T} {tag}: status gpu {np.bincount(st, minlength=5)} oracle {np.bincount(ref['status'], minlength=5)}"
              f" du {np.abs(u - ref['u'])[ok].max():.3e} dx {np.abs(x - ref['x'])[ok].max():.3e}"
              f" dobj {(np.abs(ob - ref['obj']) / np.maximum(1, np.abs(ref['obj'])))[ok].max():.3e}"
              f" iters gpu mean {out['iters'].float().mean().item():.2f} max {out['iters'].max().item()}")
    cmp("cold", out, ref)
    if warm:
        up = ref["u"]
        hp = (np.arange(B) % 3 != 0).astype(np.uint8)
        ref2 = orc.iterative_batch(ov, x0, xref, up, hp, nthreads=8)
        out2 = engine.solve_batch(v, x0, xref, up, hp)
        torch.cuda.synchronize()
        cmp("warm", out2, ref2)
    # timing
    xd = torch.from_numpy(x0).cuda(); xr = torch.from_numpy(xref).cuda()
    o = engine.solve_batch(v, xd, xr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        engine.solve_batch(v, xd, xr, out=o)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"   time/batch {dt*1e6:.1f} us -> {B/dt:.3e} solves/s")

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    for name in ["dsr_ackermann", "diff_drive", "dsr_crab", "dsr_diff", "sim_ackermann", "sim_diff", "sim_crab"]:
        run(name, B)
    run("mpc_py", 4096)
    run("dsr_ackermann", 2048, T=30)
    run("dsr_diff", 2048, T=30)
