"""The entry point that allocates nothing (mpcx_solve_batch_noalloc) against mpcx_solve_batch,
through the C ABI: same kernels on the library's fixed workspace array, so the outputs must be
bit-identical.  Two calls in a row, the second with a smaller batch (the array keeps the first
call's contents), then a batch beyond the array's size, which must be refused."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import make_problem

pytestmark = pytest.mark.gpu


def _call(entry, v, x0, xref, uprev):
    from model_predictive_control_amd import _lib
    B, T, NU = x0.shape[0], v.T, v.nu
    dev = x0.device
    out = dict(u=torch.full((B, NU, T), -7.0, dtype=torch.float64, device=dev),
               x=torch.full((B, 3, T + 1), -7.0, dtype=torch.float64, device=dev),
               obj=torch.full((B,), -7.0, dtype=torch.float64, device=dev),
               status=torch.full((B,), -7, dtype=torch.int32, device=dev),
               iters=torch.full((B,), -7, dtype=torch.int32, device=dev))
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    rc = entry(C.byref(v), B, p(x0), p(xref), p(uprev), None, p(out["u"]), p(out["x"]), p(out["obj"]),
               p(out["status"]), p(out["iters"]), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return rc, {k: t.cpu().numpy() for k, t in out.items()}


def test_noalloc_entry_point_matches_the_allocating_one():
    from model_predictive_control_amd import _lib
    lib = _lib.load()
    assert torch.cuda.is_available()
    cap = lib.mpcx_static_workspace_bytes()
    assert cap == 64 << 20
    # T = 30 beyond what the chip holds one wavefront per instance -> stage-wise kernel, which
    # keeps its per-lane arrays in the workspace
    v, ov, x0, xref = make_problem("dsr_ackermann", 2500, seed=91, T=30)
    assert lib.mpcx_workspace_bytes(C.byref(v), 2500) > 0
    d_x0, d_xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
    for n, warm in ((2500, False), (1700, True)):
        up = None
        if warm:
            rc, cold = _call(lib.mpcx_solve_batch, v, d_x0[:n], d_xref[:n], None)
            assert rc == 0
            up = torch.from_numpy(cold["u"]).cuda()
        rc, want = _call(lib.mpcx_solve_batch, v, d_x0[:n], d_xref[:n], up)
        assert rc == 0 and (want["status"] == 0).all()
        rc, got = _call(lib.mpcx_solve_batch_noalloc, v, d_x0[:n], d_xref[:n], up)
        assert rc == 0, lib.mpcx_last_error()
        for k in want:
            assert np.array_equal(got[k], want[k]), k
    # a chained mpc_sim solve at another horizon uses the same array with another layout
    v8, _, x8, r8 = make_problem("sim_ackermann", 300, seed=92, T=8)
    assert lib.mpcx_workspace_bytes(C.byref(v8), 300) > 0
    d_x8, d_r8 = torch.from_numpy(x8).cuda(), torch.from_numpy(r8).cuda()
    rc, want = _call(lib.mpcx_solve_batch, v8, d_x8, d_r8, None)
    rc2, got = _call(lib.mpcx_solve_batch_noalloc, v8, d_x8, d_r8, None)
    assert rc == 0 and rc2 == 0
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    # kernels that live in registers and LDS need no workspace: any size goes through
    v5, _, x5, r5 = make_problem("dsr_ackermann", 5000, seed=93)
    d_x5, d_r5 = torch.from_numpy(x5).cuda(), torch.from_numpy(r5).cuda()
    rc, want = _call(lib.mpcx_solve_batch, v5, d_x5, d_r5, None)
    rc2, got = _call(lib.mpcx_solve_batch_noalloc, v5, d_x5, d_r5, None)
    assert rc == 0 and rc2 == 0
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    # beyond the array: refused with a clear error, nothing launched (outputs untouched)
    per = lib.mpcx_workspace_bytes(C.byref(v), 2500) // 2560     # bytes per instance slot (batches round up to 64)
    too_many = int(cap // per) + 64
    big_x0 = d_x0[:1].expand(too_many, 3).contiguous()
    big_xref = d_xref[:1].expand(too_many, 3, 31).contiguous()
    rc, got = _call(lib.mpcx_solve_batch_noalloc, v, big_x0, big_xref, None)
    assert rc == _lib_E_NOMEM() and b"allocates nothing" in lib.mpcx_last_error()
    assert (got["status"] == -7).all() and (got["obj"] == -7.0).all()


def _lib_E_NOMEM():
    return -4


def test_noalloc_call_inside_a_graph_and_free_memory_stands_still():
    """After the documented set-up (mpcx_static_workspace_bytes() on the device, one warm-up call),
    the call can be recorded into a HIP graph and replayed bit-identically, and the device's
    free memory (hipMemGetInfo) is the same before and after a run of further calls."""
    from model_predictive_control_amd import _lib
    lib = _lib.load()
    lib.mpcx_static_workspace_bytes()
    v, ov, x0, xref = make_problem("dsr_ackermann", 1500, seed=94, T=30)
    assert lib.mpcx_workspace_bytes(C.byref(v), 1500) > 0
    d_x0, d_xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
    rc, want = _call(lib.mpcx_solve_batch_noalloc, v, d_x0, d_xref, None)       # warm-up call
    assert rc == 0 and (want["status"] == 0).all()
    B, T, NU = 1500, v.T, v.nu
    out = dict(u=torch.zeros((B, NU, T), dtype=torch.float64, device="cuda"),
               x=torch.zeros((B, 3, T + 1), dtype=torch.float64, device="cuda"),
               obj=torch.zeros((B,), dtype=torch.float64, device="cuda"),
               status=torch.zeros((B,), dtype=torch.int32, device="cuda"),
               iters=torch.zeros((B,), dtype=torch.int32, device="cuda"))
    p = lambda t: C.c_void_p(t.data_ptr())

    def launch():
        rc = lib.mpcx_solve_batch_noalloc(C.byref(v), B, p(d_x0), p(d_xref), None, None, p(out["u"]), p(out["x"]),
                                          p(out["obj"]), p(out["status"]), p(out["iters"]),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.mpcx_last_error()

    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, (free0, free1)
    for k in want:
        assert np.array_equal(out[k].cpu().numpy(), want[k]), k

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launch()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    for t in out.values():
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for k in want:
        assert np.array_equal(out[k].cpu().numpy(), want[k]), k
    free2, _ = torch.cuda.mem_get_info()
    for t in out.values():
        t.zero_()
    g.replay(); g.replay()
    torch.cuda.synchronize()
    free3, _ = torch.cuda.mem_get_info()
    assert free3 == free2, (free2, free3)
    for k in want:
        assert np.array_equal(out[k].cpu().numpy(), want[k]), k
