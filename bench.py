#!/usr/bin/env python3
"""Headline benchmark: MPC solves/s on 65,536 batched T=5 general-bicycle (dual-steer,
ackermann equalities) instances per GPU.

A "step" is one iterative_linear_mpc_control pass over the whole batch (rollout + T
linearisations + one QP per instance) with every input already resident in HBM.
One process per GPU; instances are independent, so ranks share nothing and the only
collective is the max-over-ranks of the elapsed time ("weak" scaling: 65,536 per rank).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from model_predictive_control_amd import engine, sharding, variants, workload  # noqa: E402

VARIANT = "dsr_ackermann"
BATCH = 65536
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak
SOLVE_KERNEL = "k_solve_pair5<dual-steer, single QP>"   # two lanes per instance, T=5, everything on chip
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "hbm_traffic.json")


def measured_traffic():
    """HBM bytes per launch of the solve kernel from rocprofv3 counter passes of this same
    command (FETCH_SIZE doubled as gfx950 tallies 128-B requests at 64 B, plus WRITE_SIZE);
    written by scripts/summarise_pmc.py. None when no counter pass has been recorded."""
    try:
        with open(TRAFFIC_FILE) as f:
            rec = json.load(f)
        return float(rec["bytes_per_launch"]) if rec.get("batch") == BATCH else None
    except (OSError, ValueError, KeyError):
        return None


def measured_valu_busy():
    """Fraction of the solve kernel's wave cycles that issue a VALU instruction, from the SQ
    counter passes recorded in profiles/ (scripts/pmc_pair.sh); None if not recorded."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pair5_sq_counters.json")) as f:
            return float(json.load(f)["valu_busy_frac"])
    except (OSError, ValueError, KeyError):
        return None


def algorithmic_bytes(nu: int, T: int) -> int:
    """Compulsory traffic of one solve at the float64 API: read xref 24(T+1), x0 24, uprev
    8 NU T, has_prev 1; write u 8 NU T, x 24(T+1), obj 8, status 4, iters 4."""
    return 16 * nu * T + 48 * (T + 1) + 41


WARM_TICKS = 9     # closed-loop ticks run before the timed solve: the timed pass is tick 10


def build_inputs(rank: int, B: int):
    """Synthetic vehicles on synthetic spline courses, driven for WARM_TICKS closed-loop ticks
    on the GPU; returns the inputs of the next tick's solve (device tensors): state, reference
    window, previous optimum (fed back unshifted, as the reference's loop does) and its flag."""
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
    `seconds_target` seconds of CPU work in all."""
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
    try:
        import cvxpy  # noqa: F401
        cvx = "importable (not timed: the reference's files are not on this box)"
    except Exception:
        cvx = "cvxpy baseline unavailable offline"
    return {"value": done / dt, "unit": "solves/s", "cores": threads, "kind": "port",
            "sample": f"{done} solves = {done // n} passes over the benchmark's own {n} instances, C float64 "
                      f"interior-point port of the same path, {threads} OpenMP threads, {dt:.1f} s",
            "single_thread_value": done1 / dt1, "single_thread_sample": f"{done1} solves, {dt1:.1f} s", "cvxpy": cvx}


def other_configs(steps: int = 3):
    """The survey's other single-GPU configurations, timed briefly beside the headline one
    (warm-started from their own cold optimum, inputs resident): 262,144 diff-drive T=5,
    65,536 mpc.py T=10, 65,536 dual-steer T=30."""
    res = []
    courses = workload.make_courses(64, seed=1001)
    for name, B, T, kernel in (("diff_drive", 262144, None, "k_solve_pair5<diff-drive>"),
                               ("mpc_py", 65536, None, "k_solve_pair10<dual-steer>"),
                               ("dsr_ackermann", 65536, 30, "k_solve_stage<dual-steer>")):
        v = variants.get(name, T=T) if T else variants.get(name)
        inst = workload.make_instances(courses, B, seed=2003)
        window = workload.reference_window_next_points if name == "mpc_py" else workload.reference_window
        d_xref = torch.from_numpy(window(courses, inst, v.T)).cuda()
        d_x0 = torch.from_numpy(inst["state"][:, :3].copy()).cuda()
        out = engine.solve_batch(v, d_x0, d_xref)
        up = out["u"].clone()
        engine.solve_batch(v, d_x0, d_xref, up, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            engine.solve_batch(v, d_x0, d_xref, up, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ab = algorithmic_bytes(v.nu, v.T)
        res.append({"workload": f"{name} T={v.T} batch {B}", "value": B / dt, "unit": "solves/s", "ms_per_step": dt * 1e3,
                    "kernel": kernel, "solved_fraction": float((out["status"] <= 1).float().mean().item()),
                    "algorithmic_GBps": ab * B / dt / 1e9})
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    args = ap.parse_args()

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

    B = args.batch
    v, d_x0, d_xref, d_uprev, d_has = build_inputs(rank, B)
    out = engine.solve_batch(v, d_x0, d_xref, d_uprev, d_has)
    torch.cuda.synchronize()
    ok = int((out["status"] <= 1).sum().item())
    warm_fraction = float(d_has.float().mean().item())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        engine.solve_batch(v, d_x0, d_xref, d_uprev, d_has, out=out)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k].record()                                 # same stream the kernel is launched on
        engine.solve_batch(v, d_x0, d_xref, d_uprev, d_has, out=out)
    ev[args.steps].record()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dist, "cuda")
    ok = int(sharding.sum_over_ranks(ok, dist, "cuda"))
    per_launch_ms = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)])

    if rank == 0:
        kernel_s = float(per_launch_ms.mean()) * 1e-3
        bytes_per_launch = algorithmic_bytes(v.nu, v.T) * B
        achieved = bytes_per_launch / kernel_s / 1e9
        line = {
            "metric": "MPC solves/sec (T=5 bicycle, batch 64k)",
            "value": sharding.whole_job_rate(B, world, args.steps, elapsed),
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{B} independent T=5 general-bicycle (dual-steer NX=3 NU=4, ackermann "
                                   "equalities, mpc_dsr.py:201-267) instances per GPU on 1024 synthetic cubic-spline "
                                   f"courses; the timed solve is tick {WARM_TICKS + 1} of a closed-loop run (previous optimum "
                                   "fed back unshifted)", "variant": VARIANT, "batch_per_gpu": B,
                       "horizon": v.T, "parallelism": f"independent shards x{world}, no collective"},
            "p50_batch_latency_ms": float(np.median(per_launch_ms)),
            "solved_fraction": ok / (B * world), "warm_fraction_rank0": warm_fraction,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic() if B == BATCH else None,
                         "kernel": SOLVE_KERNEL, "kernel_ms": kernel_s * 1e3,
                         "kernel_ms_note": "HIP events on the launch stream, back-to-back launches: includes the ~4 us "
                                           "dispatch gap between consecutive kernels that a kernel trace leaves out",
                         "algorithmic_bytes_per_solve": algorithmic_bytes(v.nu, v.T),
                         "valu_busy_frac": measured_valu_busy() if B == BATCH else None,
                         "note": "the solve is FP64-VALU issue bound (5 interior-point iterations + ~4 active-set "
                                 "rounds, one 10x10 Cholesky each, per instance); HBM fraction reported as the task asks"},
        }
        if not args.no_other_configs and world == 1:
            line["other_configs"] = other_configs()
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(d_x0.cpu().numpy(), d_xref.cpu().numpy(), d_uprev.cpu().numpy(),
                                                d_has.cpu().numpy())
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
