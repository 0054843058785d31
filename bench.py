#!/usr/bin/env python3
"""Headline benchmark: MPC solves/s on 65,536 batched T=5 general-bicycle (dual-steer,
ackermann equalities) instances per GPU.

A "step" is one iterative_linear_mpc_control pass over the whole batch (rollout + T
linearisations + one QP per instance) with every input already resident in HBM.
One process per GPU; instances are independent, so ranks share nothing and the only
collectives are the barrier and the max-over-ranks of the elapsed time.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus 8                      # starts 8 ranks itself (one GPU each)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N --scaling strong --workload mixed4m
        # 4,194,304 instances, a quarter each diff-drive / dual-steer ackermann / crab / fixed
        # steer (T = 5), sorted by variant and split evenly over the N GPUs; reports the solve
        # rate and, separately, the time of the per-GPU gather of the results into pinned host memory
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VARIANT = "dsr_ackermann"
BATCH = 65536
MIXED_VARIANTS = ("diff_drive", "dsr_ackermann", "dsr_crab", "dsr_diff")
MIXED_TOTAL = 4194304
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak
SOLVE_KERNEL = "k_solve_pair5<dual-steer, single QP>"   # two lanes per instance, T=5, everything on chip
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "hbm_traffic.json")
SQ_FILE = os.path.join(ROOT, "profiles", "pair5_sq_counters.json")
F64_FILE = os.path.join(ROOT, "profiles", "pair5_f64_counters.json")
WARM_TICKS = 9     # closed-loop ticks run before the timed solve: the timed pass is tick 10
# FP64 vector peak: 256 CUs x 4 SIMDs x 16 lanes/clock x 2 flops (fma) x 2.4 GHz (half the
# packed-FP32 vector rate of 157.3 TFLOP/s); MFMA plays no part in this path
FP64_VALU_PEAK_TFLOPS = 78.6
# Untimed launches before the warm-up steps: an idle MI355X needs about 200 back-to-back
# launches (20 ms) of this kernel to reach the clock it then holds (scripts/clock_ramp.py:
# 108 us for the first 50 launches, 85, 83, 81, then 79.9 us for good); a 2 ms timed region
# entered from idle measures the ramp, not the rate a running fleet sees.
SETTLE_SECONDS = 0.06


def ranks_to_spawn(gpus: int, env) -> int:
    """How many rank processes this invocation has to start itself: `--gpus N` with N > 1 and no
    launcher around it (no WORLD_SIZE in the environment); 0 otherwise."""
    if gpus > 1 and "WORLD_SIZE" not in env:
        return gpus
    return 0


def kernel_source_hash() -> str:
    """Identifies the kernels a recorded profile belongs to: sha256 over csrc/ and the header."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "model_predictive_control_amd", "csrc")
    for f in sorted(os.listdir(base)) + [os.path.join(ROOT, "include", "mpcx.h")]:
        with open(f if os.path.isabs(f) else os.path.join(base, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def recorded_counters(batch: int):
    """Counter results recorded under profiles/ by separate rocprofv3 --pmc passes of this same
    command (scripts/gpu_validate.sh): HBM bytes per launch of the solve kernel (FETCH_SIZE
    doubled as gfx950 tallies 128-B requests at 64 B, plus WRITE_SIZE) and the fraction of wave
    cycles that issue a VALU instruction.  They are replays of files, not measurements of this
    run, and are dropped when the kernels have changed since they were recorded."""
    out = {"traffic": None, "valu_busy_frac": None, "f64": None, "from_profiles": None}
    now = kernel_source_hash()
    src = {}
    try:
        with open(TRAFFIC_FILE) as f:
            rec = json.load(f)
        if rec.get("batch") == batch and rec.get("csrc_sha") == now:
            out["traffic"] = float(rec["bytes_per_launch"])
            src["traffic"] = {"file": os.path.relpath(TRAFFIC_FILE, ROOT), "csrc_sha": rec["csrc_sha"]}
    except (OSError, ValueError, KeyError):
        pass
    try:
        with open(SQ_FILE) as f:
            rec = json.load(f)
        if rec.get("batch") == batch and rec.get("csrc_sha") == now:
            out["valu_busy_frac"] = float(rec["valu_busy_frac"])
            src["valu_busy_frac"] = {"file": os.path.relpath(SQ_FILE, ROOT), "csrc_sha": rec["csrc_sha"]}
    except (OSError, ValueError, KeyError):
        pass
    try:
        with open(F64_FILE) as f:
            rec = json.load(f)
        if rec.get("batch") == batch and rec.get("csrc_sha") == now:
            out["f64"] = rec
            src["roofline_valu"] = {"file": os.path.relpath(F64_FILE, ROOT), "csrc_sha": rec["csrc_sha"]}
    except (OSError, ValueError, KeyError):
        pass
    if src:
        out["from_profiles"] = src
    return out


def settle(step, seconds: float = SETTLE_SECONDS):
    """Untimed back-to-back launches until the GPU holds its clock (see SETTLE_SECONDS)."""
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        n += 50
    return n


def algorithmic_bytes(nu: int, T: int) -> int:
    """Compulsory traffic of one solve at the float64 API: read xref 24(T+1), x0 24, uprev
    8 NU T, has_prev 1; write u 8 NU T, x 24(T+1), obj 8, status 4, iters 4."""
    return 16 * nu * T + 48 * (T + 1) + 41


def build_inputs(rank: int, B: int):
    """Synthetic vehicles on synthetic spline courses, driven for WARM_TICKS closed-loop ticks
    on the GPU; returns the inputs of the next tick's solve (device tensors): state, reference
    window, previous optimum (fed back unshifted, as the reference's loop does) and its flag."""
    import torch
    from model_predictive_control_amd import engine, sharding, variants, workload
    v = variants.get(VARIANT)
    courses = workload.make_courses(1024, seed=1001)
    inst = workload.make_instances(courses, B, seed=sharding.rank_seed(2002, rank))
    table = engine.CourseTable(courses)
    loop = engine.ClosedLoop(v, table, inst["state"][:, :4].copy(), inst["course_id"], pind0=inst["pind"])
    loop.run(WARM_TICKS)
    loop.prepare()
    torch.cuda.synchronize()
    return v, loop.x0.clone(), loop.xref.clone(), loop.uprev.clone(), loop.has_prev.clone()


def cpu_baseline(x0, xref, uprev, has_prev, seconds_target: float = 12.0):
    """The C restatement of the same path (oracle/mpc_oracle.c, polish off) timed on the
    host cores of this box over repeated passes of the benchmark's own batch, about
    `seconds_target` seconds of CPU work in all; plus, only where cvxpy is importable, the
    reference's own formulation (cvxpy + ECOS, rebuilt per call as the reference does) on one core."""
    from oracle import mpc_oracle as O
    from oracle import orc
    ov = O.variant(VARIANT)
    orc.set_polish(False)
    try:
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        threads = max(1, min(16, os.cpu_count() or 1))
    n = x0.shape[0]
    orc.iterative_batch(ov, x0[:4096], xref[:4096], uprev[:4096], has_prev[:4096], nthreads=threads)   # warm the pool
    done, t0 = 0, time.perf_counter()
    while True:
        orc.iterative_batch(ov, x0, xref, uprev, has_prev, nthreads=threads)
        done += n
        dt = time.perf_counter() - t0
        if dt >= seconds_target * 0.75:
            break
    n1 = min(n, 16384)
    done1, t1 = 0, time.perf_counter()
    while True:
        orc.iterative_batch(ov, x0[:n1], xref[:n1], uprev[:n1], has_prev[:n1], nthreads=1)
        done1 += n1
        dt1 = time.perf_counter() - t1
        if dt1 >= seconds_target * 0.25:
            break
    orc.set_polish(True)
    from tools import cvxpy_baseline
    cvx = cvxpy_baseline.run(VARIANT, x0, xref, uprev, has_prev, n_solves=200)
    return {"value": done / dt, "unit": "solves/s", "cores": threads, "kind": "port",
            "sample": f"{done} solves = {done // n} passes over the benchmark's own {n} instances, C float64 "
                      f"interior-point port of the same path, {threads} OpenMP threads, {dt:.1f} s",
            "single_thread_value": done1 / dt1, "single_thread_sample": f"{done1} solves, {dt1:.1f} s", "cvxpy": cvx}


def other_configs(steps: int = 3):
    """The survey's other single-GPU configurations, timed briefly beside the headline one
    (inputs resident; "warm" = linearised at their own cold optimum, as the reference's loop
    feeds the previous answer back): 65,536 and 262,144 diff-drive T=5, 65,536 mpc.py T=10, and mpc_dsr.py
    at T=30 both ways: the QP its iterative_linear_mpc_control actually calls (fixed steer, stage
    weights Q 2^t, mpc_dsr.py:193,349) and the ackermann one, at 65,536 instances (throughput) and at
    1,024 and 1 (latency: what a small fleet or the reference's single vehicle waits per solve)."""
    import torch
    from model_predictive_control_amd import engine, variants, workload
    res = []
    courses = workload.make_courses(64, seed=1001)
    for name, B, T, kernel in (("diff_drive", 65536, None, "k_solve_pair5<diff-drive>"),
                               ("diff_drive", 262144, None, "k_solve_pair5<diff-drive>"),
                               ("sim_ackermann", 65536, None, "k_solve_pair5<dual-steer, chain of 5 QPs> (MPC.iterativeLMPC, mpc_sim.py:284-310)"),
                               ("mpc_py", 65536, None, "k_solve_quad<dual-steer, T=10> (four lanes per instance)"),
                               ("mpc_py", 32768, None, "k_solve_quad<dual-steer, T=10> (four lanes per instance)"),
                               ("dsr_diff", 65536, 30, "k_solve_stage<dual-steer>"),
                               ("dsr_ackermann", 65536, 30, "k_solve_stage<dual-steer>"),
                               ("dsr_diff", 1024, 30, "k_solve_wave<dual-steer, T=30> (one wavefront per instance) + k_solve_stage<REDO>"),
                               ("dsr_ackermann", 1024, 30, "k_solve_wave<dual-steer, T=30> (one wavefront per instance) + k_solve_stage<REDO>"),
                               ("dsr_ackermann", 8192, 30, "k_solve_wave<dual-steer, T=30> (the largest batch routed to it) + k_solve_stage<REDO>"),
                               ("dsr_diff", 1, 30, "k_solve_wave<dual-steer, T=30> (one wavefront per instance) + k_solve_stage<REDO>")):
        v = variants.get(name, T=T) if T else variants.get(name)
        inst = workload.make_instances(courses, B, seed=2003)
        window = workload.reference_window_next_points if name == "mpc_py" else workload.reference_window
        d_xref = torch.from_numpy(window(courses, inst, v.T)).cuda()
        d_x0 = torch.from_numpy(inst["state"][:, :3].copy()).cuda()
        out = engine.solve_batch(v, d_x0, d_xref)
        up = out["u"].clone()
        ab = algorithmic_bytes(v.nu, v.T)
        for start, u in (("cold", None), ("warm", up)):
            if start == "cold" and T != 30:
                continue
            n = steps if v.T > 10 and B > 16384 else (4 * steps if v.T > 10 and B > 4096 else 40 * steps)      # the short kernels need more launches for a stable figure
            for _ in range(max(1, n // 4)):
                engine.solve_batch(v, d_x0, d_xref, u, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                engine.solve_batch(v, d_x0, d_xref, u, out=out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            res.append({"workload": f"{name} T={v.T} batch {B} {start}", "value": B / dt, "unit": "solves/s",
                        "ms_per_step": dt * 1e3, "kernel": kernel,
                        "solved_fraction": float((out["status"] <= 1).float().mean().item()),
                        "failed": int((out["status"] == 4).sum().item()),
                        "algorithmic_GBps": ab * B / dt / 1e9})
    res.append(closed_loop_tick())
    res.append(mixed_sweep_one_gpu())
    return res


def fused_tick_bytes(nu: int, T: int) -> int:
    """Compulsory traffic of one fused closed-loop tick per vehicle: the iterative entry's bytes with
    the 24 (T+1)-byte reference window read replaced by state 32 B + previous index 4 B + course id
    4 B + ten course points and the end point 176 B, plus the new index 4 B written."""
    return algorithmic_bytes(nu, T) - 24 * (T + 1) + 32 + 4 + 4 + 176 + 4


def mixed_sweep_one_gpu(total: int = MIXED_TOTAL, steps: int = 3):
    """The 4 x 1,048,576 mixed-variant sweep on this one GPU (the strong-scaling workload at N = 1):
    four launches per step; the gather of the results into pinned host memory timed beside it."""
    import torch
    from model_predictive_control_amd import engine
    parts = mixed_inputs(0, 1, total)

    def step():
        for p in parts:
            engine.solve_batch(p["v"], p["x0"], p["xref"], p["uprev"], out=p["out"])
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    host = [{k: torch.empty(t.shape, dtype=t.dtype).pin_memory() for k, t in p["out"].items()} for p in parts]
    torch.cuda.synchronize()
    tg = time.perf_counter()
    for p, h in zip(parts, host):
        for k, t in p["out"].items():
            h[k].copy_(t, non_blocking=True)
    torch.cuda.synchronize()
    gather_s = time.perf_counter() - tg
    n = sum(p["n"] for p in parts)
    nbytes = sum(t.numel() * t.element_size() for h in host for t in h.values())
    ab = sum(algorithmic_bytes(p["v"].nu, p["v"].T) * p["n"] for p in parts)
    return {"workload": f"mixed sweep {n} instances ({', '.join(MIXED_VARIANTS)}; T=5), one GPU", "value": n / dt, "unit": "solves/s",
            "ms_per_step": dt * 1e3, "kernel": "k_solve_pair5, four instantiations, one launch per variant",
            "solved_fraction": sum(int((p["out"]["status"] <= 1).sum().item()) for p in parts) / n,
            "failed": sum(int((p["out"]["status"] == 4).sum().item()) for p in parts),
            "algorithmic_GBps": ab / dt / 1e9,
            "gather": {"ms": gather_s * 1e3, "bytes": nbytes, "GBps": nbytes / gather_s / 1e9,
                       "how": "u, x, obj, status, iters copied into pinned host memory; not part of value"}}


def closed_loop_tick(B: int = BATCH, n: int = 200):
    """The fused tick (reference window from the course tables on the device -> solve -> state
    advance and goal test, no host round trip) of a fleet of the headline variant: per vehicle
    and tick it reads the state (32 B), previous index and course id (8 B), ten course points
    and the end point (176 B), the warm start (16 NU T B) and writes the controls, states,
    objective, status, the new index and state."""
    import torch
    from model_predictive_control_amd import engine, variants, workload
    v = variants.get(VARIANT)
    courses = workload.make_courses(1024, seed=1001)
    inst = workload.make_instances(courses, B, seed=2003)
    loop = engine.ClosedLoop(v, engine.CourseTable(courses), inst["state"][:, :4].copy(), inst["course_id"], pind0=inst["pind"])
    loop.run(100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.run(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return {"workload": f"closed-loop tick {VARIANT} T={v.T} fleet {B}", "value": B / dt, "unit": "vehicle-ticks/s",
            "ms_per_step": dt * 1e3, "kernel": "k_nearest_ref + k_solve_pair5 + k_advance",
            "algorithmic_bytes_per_vehicle_tick": fused_tick_bytes(v.nu, v.T),
            "algorithmic_GBps": fused_tick_bytes(v.nu, v.T) * B / dt / 1e9,
            "still_driving": int((loop.done == 0).sum().item()), "failed": int((loop.out["status"] == 4).sum().item())}


def mixed_inputs(rank: int, world: int, total: int):
    """This rank's part of the variant-sorted mixed batch: [(variant struct, x0, xref, uprev)] per
    variant piece, device tensors, warm-started from the piece's own cold optimum."""
    import numpy as np
    import torch
    from model_predictive_control_amd import engine, sharding, variants, workload
    counts = [total // len(MIXED_VARIANTS)] * len(MIXED_VARIANTS)
    counts[-1] += total - sum(counts)
    courses = workload.make_courses(1024, seed=1001)
    table = engine.CourseTable(courses)
    parts = []
    for k, lo, hi in sharding.variant_segments(counts, rank, world):
        v = variants.get(MIXED_VARIANTS[k])
        inst = workload.make_instances(courses, counts[k], seed=3003 + k)      # the whole block, then this rank's rows
        sl = slice(lo, hi)
        state = np.ascontiguousarray(inst["state"][sl])
        _, _, xref = engine.nearest_ref_batch(1, v.T, v.dt, 2, state, inst["course_id"][sl], inst["pind"][sl], table)
        x0 = torch.from_numpy(np.ascontiguousarray(state[:, :3])).cuda()
        out = engine.solve_batch(v, x0, xref)
        parts.append(dict(name=MIXED_VARIANTS[k], v=v, x0=x0, xref=xref, uprev=out["u"].clone(), out=out, n=hi - lo))
    torch.cuda.synchronize()
    return parts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--workload", choices=("headline", "mixed4m"), default=None)
    ap.add_argument("--total", type=int, default=MIXED_TOTAL, help="instances of the mixed workload (all ranks together)")
    ap.add_argument("--gather", choices=("host", "device"), default="host",
                    help="mixed4m: where the results are gathered: each GPU into pinned host memory (default), or "
                         "every GPU's controls and statuses onto every GPU by an all-gather over RCCL")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    args = ap.parse_args()
    workload_name = args.workload or ("mixed4m" if args.scaling == "strong" else "headline")
    if workload_name == "mixed4m":
        args.scaling = "strong"
    if args.steps is None:
        args.steps = 50 if workload_name == "headline" else 10
    if args.warmup is None:
        args.warmup = 5 if workload_name == "headline" else 2

    # `--gpus N` without a launcher: start the N ranks here, before anything touches the GPU
    n_spawn = ranks_to_spawn(args.gpus, os.environ)
    if n_spawn:
        from model_predictive_control_amd import sharding
        sys.exit(sharding.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], n_spawn))

    import numpy as np
    import torch
    from model_predictive_control_amd import engine, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    # one GPU per rank; MPCX_BENCH_BACKEND=gloo lets several ranks rehearse the multi-process
    # flow on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device)
    backend = os.environ.get("MPCX_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if workload_name == "headline":
        B = args.batch
        v, d_x0, d_xref, d_uprev, d_has = build_inputs(rank, B)
        out = engine.solve_batch(v, d_x0, d_xref, d_uprev, d_has)
        torch.cuda.synchronize()
        ok = int((out["status"] <= 1).sum().item())
        warm_fraction = float(d_has.float().mean().item())

        def step():
            engine.solve_batch(v, d_x0, d_xref, d_uprev, d_has, out=out)
        units_here, bytes_here = B, algorithmic_bytes(v.nu, v.T) * B
    else:
        parts = mixed_inputs(rank, world, args.total)

        def step():
            for p in parts:
                engine.solve_batch(p["v"], p["x0"], p["xref"], p["uprev"], out=p["out"])
        step()
        torch.cuda.synchronize()
        ok = sum(int((p["out"]["status"] <= 1).sum().item()) for p in parts)
        units_here = sum(p["n"] for p in parts)
        bytes_here = sum(algorithmic_bytes(p["v"].nu, p["v"].T) * p["n"] for p in parts)

    settled = settle(step)
    for _ in range(args.warmup):
        step()
    # HIP events on the launch stream bracket the K steps (one before the first launch, one after
    # the last: an event between every two launches costs about 6 us of dispatch each)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dist, red_dev)
    ok = int(sharding.sum_over_ranks(ok, dist, red_dev))
    units_all = int(sharding.sum_over_ranks(units_here, dist, red_dev))
    kernel_s = ev0.elapsed_time(ev1) * 1e-3 / args.steps
    # latency of one batch call, outside the timed region: every step bracketed by its own events
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for k in range(args.steps):
        ev[k].record()
        step()
    ev[args.steps].record()
    torch.cuda.synchronize()
    per_step_ms = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)])

    gather_ms = None
    if workload_name == "mixed4m":
        # the final result gather: each GPU copies its outputs into pinned host memory (no
        # GPU-to-GPU traffic); timed on its own, never part of the solve rate
        host = [{k: torch.empty(t.shape, dtype=t.dtype).pin_memory() for k, t in p["out"].items()} for p in parts]
        barrier()
        tg = time.perf_counter()
        for p, h in zip(parts, host):
            for k, t in p["out"].items():
                h[k].copy_(t, non_blocking=True)
        torch.cuda.synchronize()
        gather_ms = sharding.max_over_ranks((time.perf_counter() - tg) * 1e3, dist, red_dev)
        gathered_bytes = sharding.sum_over_ranks(sum(t.numel() * t.element_size() for h in host for t in h.values()), dist, red_dev)
        device_gather = None
        if args.gather == "device" and dist is not None and backend == "nccl":
            # device-resident alternative: controls and statuses of every rank on every GPU (what a fleet
            # controller that stays on the GPUs needs), one all-gather per array over xGMI
            u_mine = torch.cat([p["out"]["u"][:, :2].reshape(p["n"], -1) for p in parts], dim=0)     # first two rows: every variant has them
            st_mine = torch.cat([p["out"]["status"] for p in parts], dim=0)
            counts = [hi - lo for lo, hi in sharding.partition(args.total, world)]
            barrier()
            td = time.perf_counter()
            u_all = sharding.all_gather_rows(u_mine, counts, dist)
            st_all = sharding.all_gather_rows(st_mine, counts, dist)
            torch.cuda.synchronize()
            dms = sharding.max_over_ranks((time.perf_counter() - td) * 1e3, dist, red_dev)
            device_gather = {"ms": dms, "bytes_per_gpu": int(u_all.numel() * 8 + st_all.numel() * 4),
                             "how": "all-gather of the first two control rows and the statuses onto every GPU (RCCL)"}

    if rank == 0:
        achieved = bytes_here / kernel_s / 1e9
        line = {
            "metric": "MPC solves/sec (T=5 bicycle, batch 64k)",
            "value": units_all * args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "solved_fraction": ok / units_all,
            "settle_launches": settled,
            "settle_note": f"{settled} untimed steps ({SETTLE_SECONDS * 1e3:.0f} ms) before the warm-up steps bring the GPU from idle "
                           "to the clock it holds under this load (scripts/clock_ramp.py)",
        }
        if workload_name == "headline":
            line["config"] = {"workload": f"{B} independent T=5 general-bicycle (dual-steer NX=3 NU=4, ackermann "
                                          "equalities, mpc_dsr.py:201-267) instances per GPU on 1024 synthetic cubic-spline "
                                          f"courses; the timed solve is tick {WARM_TICKS + 1} of a closed-loop run (previous "
                                          "optimum fed back unshifted)", "variant": VARIANT, "batch_per_gpu": B,
                              "horizon": v.T, "parallelism": f"independent shards x{world}, no collective"}
            line["p50_batch_latency_ms"] = float(np.median(per_step_ms))
            line["warm_fraction_rank0"] = warm_fraction
            rec = recorded_counters(B)
            line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": achieved / HBM_PEAK_GBS, "traffic": rec["traffic"],
                                "kernel": SOLVE_KERNEL, "kernel_ms": kernel_s * 1e3,
                                "kernel_ms_note": "HIP events on the launch stream around the K back-to-back launches of the timed "
                                                  "region, divided by K: includes the dispatch gap between consecutive kernels "
                                                  "that a kernel trace leaves out",
                                "algorithmic_bytes_per_solve": algorithmic_bytes(v.nu, v.T),
                                "valu_busy_frac": rec["valu_busy_frac"],
                                "from_profiles": rec["from_profiles"],
                                "from_profiles_note": "traffic and valu_busy_frac are replayed from counter passes recorded "
                                                      "under profiles/ (not measured by this run) and are null when the "
                                                      "kernel sources have changed since",
                                "note": "the solve is FP64-VALU issue bound (5 interior-point iterations + ~4 active-set "
                                        "rounds, one 10x10 Cholesky each, per instance); HBM fraction reported as the task asks"}
            f64 = rec["f64"]
            if f64 is not None:
                # flops of one launch from the recorded FP64 instruction counts (every wave instruction
                # counted at 64 lanes: an upper bound where lanes are masked off) over THIS run's kernel time
                tf = float(f64["flops_per_launch"]) / kernel_s / 1e12
                line["roofline_valu"] = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": tf / FP64_VALU_PEAK_TFLOPS, "flops_per_launch": f64["flops_per_launch"],
                                         "flops_per_solve": f64["flops_per_launch"] / B,
                                         "f64_instructions_per_wave": f64["per_wave"],
                                         "note": "flops = 64 x (2 FMA + MUL + ADD + TRANS) FP64 wave instructions of one launch, from "
                                                 "a recorded SQ_INSTS_VALU_*_F64 counter pass (replayed, see roofline.from_profiles); "
                                                 "time = this run's HIP-event kernel time"}
        else:
            line["metric"] = "MPC solves/sec (T=5 mixed-model sweep, 4M instances)"
            line["config"] = {"workload": f"{units_all} instances, a quarter each of {', '.join(MIXED_VARIANTS)} (T=5), sorted "
                                          "by variant so every wavefront is homogeneous, contiguous ranges per GPU; "
                                          "warm-started from their own cold optimum", "total": units_all,
                              "parallelism": f"variant-sorted batch split x{world}, no collective"}
            if device_gather is not None:
                line["gather_device"] = device_gather
            line["gather"] = {"ms": gather_ms, "bytes": gathered_bytes,
                              "how": "per-GPU copy of u, x, obj, status, iters into pinned host memory, all ranks at "
                                     "once, slowest rank's time; not part of value"}
            line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                "kernel": "k_solve_pair5 (four instantiations, one launch per variant piece)",
                                "kernel_ms": kernel_s * 1e3, "note": "rank 0's algorithmic bytes per step / its step time"}
        if not args.no_other_configs and world == 1 and workload_name == "headline":
            line["other_configs"] = other_configs()
        if not args.no_cpu_baseline and world == 1 and workload_name == "headline":
            line["cpu_baseline"] = cpu_baseline(d_x0.cpu().numpy(), d_xref.cpu().numpy(), d_uprev.cpu().numpy(),
                                                d_has.cpu().numpy())
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
