"""N > 1 path on CPU: two gloo ranks split a batch of MPC instances, each solves its slice
(here with the C oracle standing in for the GPU engine, which tests may do), and the
gathered result equals the unsharded solve; timing reductions take the slowest rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from model_predictive_control_amd import sharding  # noqa: E402


def test_partition_is_a_balanced_cover():
    for total in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            parts = sharding.partition(total, world)
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.partition(4, 0)


def test_rank_seeds_differ():
    assert len({sharding.rank_seed(2002, r) for r in range(8)}) == 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from model_predictive_control_amd import workload
        from oracle import mpc_oracle as O, orc
        ov = O.variant("dsr_ackermann")
        courses = workload.make_courses(4, seed=11)
        inst = workload.make_instances(courses, total, seed=12)
        xref = workload.reference_window(courses, inst, ov.T)
        x0 = inst["state"][:, :3].copy()
        mine = sharding.shard_arrays({"x0": x0, "xref": xref}, rank, world)
        lo, hi = sharding.partition(total, world)[rank]
        assert mine["x0"].shape[0] == hi - lo
        res = orc.iterative_batch(ov, mine["x0"], mine["xref"], None, None, nthreads=1)
        u_all = sharding.gather_rows(res["u"], total, dist)
        # the device-resident flavour of the gather (here on CPU tensors over gloo): every rank gets all rows
        import torch
        counts = [b - a for a, b in sharding.partition(total, world)]
        everywhere = sharding.all_gather_rows(torch.from_numpy(res["u"]), counts, dist)
        assert everywhere.shape[0] == total
        lo_, hi_ = sharding.partition(total, world)[rank]
        assert np.array_equal(everywhere[lo_:hi_].numpy(), res["u"])
        np.save(os.path.join(outdir, f"allgather_{rank}.npy"), everywhere.numpy())
        # the slowest rank sets the clock; every rank sees the same number
        slow = sharding.max_over_ranks(1.0 + rank, dist)
        assert slow == float(world)
        assert sharding.sum_over_ranks(float(hi - lo), dist) == float(total)
        rate = sharding.whole_job_rate(hi - lo, world, 3, slow)
        assert rate == pytest.approx(world * (hi - lo) * 3 / world)
        if rank == 0:
            ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=1)
            np.testing.assert_array_equal(u_all, ref["u"])
            open(os.path.join(outdir, "ok"), "w").write("ok")
        else:
            assert u_all is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_split_solve_gather(tmp_path):
    world, total = 2, 37     # odd on purpose: ragged split 19 + 18
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


def test_variant_segments_cover_a_variant_sorted_batch():
    counts = [1048576] * 4
    for world in (1, 2, 3, 4, 8):
        seen = [0] * 4
        for r in range(world):
            segs = sharding.variant_segments(counts, r, world)
            assert 1 <= len(segs) <= 2 or world == 1
            for k, a, b in segs:
                assert 0 <= a < b <= counts[k]
                seen[k] += b - a
        assert seen == counts
    assert sharding.variant_segments([5, 0, 7], 1, 2) == [(2, 1, 7)]


def test_bench_launcher_spawns_the_ranks_it_is_asked_for(tmp_path):
    """`bench.py --gpus N` without a launcher around it must start N ranks itself: the spawner
    gives each child its rank, the world size and a common rendezvous address."""
    code = ("import os, sys; open(os.path.join(sys.argv[1], 'r' + os.environ['RANK']), 'w').write("
            "os.environ['WORLD_SIZE'] + ' ' + os.environ['LOCAL_RANK'] + ' ' + os.environ['MASTER_ADDR'] + ' ' + os.environ['MASTER_PORT'])")
    rc = sharding.spawn_ranks([sys.executable, "-c", code, str(tmp_path)], 3)
    assert rc == 0
    got = [open(tmp_path / f"r{r}").read().split() for r in range(3)]
    assert [g[0] for g in got] == ["3"] * 3 and [g[1] for g in got] == ["0", "1", "2"]
    assert len({g[3] for g in got}) == 1 and got[0][2] == "127.0.0.1"
    assert sharding.spawn_ranks([sys.executable, "-c", "import sys; sys.exit(3)"], 2) == 3


def test_bench_py_decides_to_spawn_before_touching_the_gpu():
    """bench.py's launcher decision is pure argument/environment logic (no torch.cuda call)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.ranks_to_spawn(8, {}) == 8
    assert bench.ranks_to_spawn(1, {}) == 0
    assert bench.ranks_to_spawn(8, {"WORLD_SIZE": "8", "RANK": "3"}) == 0     # already under a launcher


def test_bench_byte_formulas_match_the_stated_figures():
    """Algorithmic bytes per solve and per fused tick: 16 NU T + 48 (T+1) + 41, and the same with the
    reference-window read replaced by state, index, course id and course points."""
    import bench
    assert bench.algorithmic_bytes(2, 5) == 489 and bench.algorithmic_bytes(4, 5) == 649
    assert bench.algorithmic_bytes(4, 10) == 1209 and bench.algorithmic_bytes(4, 30) == 3449
    assert bench.fused_tick_bytes(2, 5) == 565 and bench.fused_tick_bytes(4, 5) == 725
