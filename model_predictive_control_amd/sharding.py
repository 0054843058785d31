"""Multi-GPU layout of a batch of MPC instances: one process per GPU, each owning a
contiguous slice of the instances.  Instances never interact, so the data path has no
collective; torch.distributed is used only to agree on timing (max over ranks) and, when
a caller wants the whole result on one rank, for a plain gather.

Nothing here touches a GPU API, so the same code runs under the `gloo` backend on CPU."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def partition(total: int, world: int) -> List[Tuple[int, int]]:
    """Balanced contiguous [start, stop) ranges: the first `total % world` ranks get one more."""
    if world < 1 or total < 0:
        raise ValueError("partition needs world >= 1 and total >= 0")
    base, extra = divmod(total, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def my_slice(total: int, rank: int, world: int) -> slice:
    lo, hi = partition(total, world)[rank]
    return slice(lo, hi)


def shard_arrays(arrays: Dict[str, np.ndarray], rank: int, world: int) -> Dict[str, np.ndarray]:
    """This rank's rows of every per-instance array (strong scaling: the batch is split)."""
    sizes = {a.shape[0] for a in arrays.values()}
    if len(sizes) != 1:
        raise ValueError("all arrays must have the same leading (instance) dimension")
    sl = my_slice(sizes.pop(), rank, world)
    return {k: a[sl] for k, a in arrays.items()}


def rank_seed(base: int, rank: int) -> int:
    """Seed of a rank's own synthetic shard (weak scaling: every rank builds a full batch)."""
    return base + 1 + 1000 * rank


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """The slowest rank's time: what a whole-job throughput must be computed from."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_rows(local: np.ndarray, total: int, dist=None) -> np.ndarray | None:
    """Rows of every rank back in instance order on rank 0 (None elsewhere).  Not part of
    the hot path: a caller that keeps results on the GPUs never needs it."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    parts: Sequence = [None] * world
    dist.gather_object(local, parts if rank == 0 else None, dst=0)
    if rank != 0:
        return None
    out = np.concatenate(list(parts), axis=0)
    if out.shape[0] != total:
        raise RuntimeError(f"gathered {out.shape[0]} rows, expected {total}")
    return out


def all_gather_rows(local, counts: Sequence[int], dist=None):
    """Device-resident alternative to the copy into host memory: every rank ends up with the rows
    of all ranks, in rank order, in its own memory (one all-gather per array over RCCL / xGMI under
    the nccl backend; gloo on CPU tensors).  counts[r] = rows of rank r; shards are padded to the
    largest so that every rank contributes the same shape.  local: a torch tensor [counts[rank], ...]."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if len(counts) != world or local.shape[0] != counts[rank]:
        raise ValueError("counts must list every rank's rows")
    most = max(counts)
    mine = local
    if local.shape[0] < most:
        pad = torch.zeros((most - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        mine = torch.cat([local, pad], dim=0)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine.contiguous())
    return torch.cat([p[:n] for p, n in zip(parts, counts)], dim=0)


def variant_segments(counts: Sequence[int], rank: int, world: int) -> List[Tuple[int, int, int]]:
    """A batch of several variants laid out variant after variant (so that every wavefront is
    homogeneous), counts[k] instances of variant k, is split into `world` contiguous ranges; this
    rank's range as (variant, start, stop) pieces with start/stop inside that variant's block."""
    total = int(sum(counts))
    lo, hi = partition(total, world)[rank]
    out, base = [], 0
    for k, n in enumerate(counts):
        a, b = max(lo, base), min(hi, base + n)
        if a < b:
            out.append((k, a - base, b - base))
        base += n
    return out


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(cmd: Sequence[str], world: int, env=None, timeout=900.0) -> int:
    """Starts `world` fresh processes of `cmd`, one rank each (RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR = 127.0.0.1 and a free MASTER_PORT in their environment, the way
    torch.distributed.run does) and polls them all.  The first rank that exits with an error ends
    the job: its siblings, which would otherwise sit in the rendezvous or a barrier until the
    collective's own timeout, are terminated (then killed) and that exit code is returned; so is
    124 once `timeout` seconds (None = no limit) have passed.  Otherwise 0 when every rank has
    ended.  The caller must not have touched the GPU: the children are new programs, nothing is
    inherited or replaced."""
    import os
    import subprocess
    import time
    port = free_port()
    procs = []
    for r in range(world):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(list(cmd), env=e))

    def stop_all():
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        t_end = time.monotonic() + 10.0
        for pr in procs:
            try:
                pr.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()

    t0 = time.monotonic()
    while True:
        codes = [pr.poll() for pr in procs]
        bad = [abs(c) for c in codes if c not in (None, 0)]
        if bad:
            stop_all()
            return bad[0]
        if all(c == 0 for c in codes):
            return 0
        if timeout is not None and time.monotonic() - t0 > timeout:
            stop_all()
            return 124
        time.sleep(0.05)


def whole_job_rate(units_per_rank: int, world: int, steps: int, elapsed_max: float) -> float:
    """Aggregate units/s over all ranks for `steps` passes timed at the slowest rank."""
    return world * units_per_rank * steps / elapsed_max
