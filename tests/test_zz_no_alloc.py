"""ilqr_solve_batch_static (the entry point that allocates nothing: workspace in a device array inside the library)
against ilqr_solve_batch on the same inputs, through the C ABI: identical results, bit for bit, on two calls in a
row, the second with a smaller batch; a batch whose workspace does not fit is refused."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve_both(B, seed):
    import torch
    import bench
    from ilqr_amd import _lib
    from ilqr_amd.batch import BatchSolver, _ptr
    dev = torch.device("cuda:0")
    s = BatchSolver(0, [2.0, 2.0], 40, 50, 20, 1e-3, device=dev)
    x0, u0 = bench.car_inputs(B, seed)
    x0s = torch.as_tensor(x0, device=dev).t().contiguous()
    us = torch.as_tensor(u0, device=dev).permute(1, 2, 0).contiguous()
    res = []
    for static in (False, True):
        u = us.clone()
        x = torch.full((40, 5, B), float("nan"), dtype=torch.float64, device=dev)
        out = s.alloc_outputs(B)
        pr = s._problem(B, B)
        oo = _lib.Outputs(*[_ptr(out.get(k)) for k, _ in _lib.Outputs._fields_])
        if static:
            assert int(s.lib.ilqr_workspace_bytes(C.byref(pr))) <= _lib.STATIC_WORKSPACE_BYTES
            s._check(s.lib.ilqr_solve_batch_static(C.byref(pr), _ptr(x0s), _ptr(u), _ptr(x), C.byref(oo), s._stream()))
        else:
            nbytes = int(s.lib.ilqr_workspace_bytes(C.byref(pr)))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            s._check(s.lib.ilqr_solve_batch(C.byref(pr), _ptr(x0s), _ptr(u), _ptr(x), C.byref(oo), _ptr(ws), nbytes,
                                            s._stream()))
        torch.cuda.synchronize()
        res.append(dict(x=x.cpu().numpy(), u=u.cpu().numpy(), **{k: v.cpu().numpy() for k, v in out.items()}))
    return res


def test_static_workspace_solve_equals_the_allocating_one():
    import torch
    assert torch.cuda.is_available(), "this test needs a GPU"
    for B, seed in ((2500, 1234), (700, 99)):      # the second call is smaller and meets the first one's leftovers
        a, b = _solve_both(B, seed)
        assert a["iters"].max() > 5 and (a["regu_rounds"] > 0).any() and (a["ls_total"] > a["iters"]).any()
        for k in a:
            assert np.array_equal(a[k], b[k], equal_nan=True), (B, k)


def test_static_workspace_refuses_what_does_not_fit():
    import torch
    assert torch.cuda.is_available(), "this test needs a GPU"
    from ilqr_amd import _lib
    from ilqr_amd.batch import BatchSolver, _ptr
    dev = torch.device("cuda:0")
    s = BatchSolver(0, [2.0, 2.0], 40, 50, 20, 1e-3, device=dev)
    B = 8192
    pr = s._problem(B, B)
    assert int(s.lib.ilqr_workspace_bytes(C.byref(pr))) > _lib.STATIC_WORKSPACE_BYTES
    x0 = torch.zeros((5, B), dtype=torch.float64, device=dev)
    u = torch.zeros((39, 2, B), dtype=torch.float64, device=dev)
    x = torch.zeros((40, 5, B), dtype=torch.float64, device=dev)
    rc = s.lib.ilqr_solve_batch_static(C.byref(pr), _ptr(x0), _ptr(u), _ptr(x), None, s._stream())
    assert rc == -2 and b"ILQR_STATIC_WORKSPACE_BYTES" in s.lib.ilqr_last_error()
    pr = s._problem(64, 64)
    pr.flags |= _lib.FLAG_TIMING
    assert s.lib.ilqr_solve_batch_static(C.byref(pr), _ptr(x0), _ptr(u), _ptr(x), None, s._stream()) == -1


def test_static_workspace_call_leaves_device_memory_as_it_found_it():
    """What the entry point exists for: after a first call (which loads the kernels and takes the address of the
    library's array) a second call changes the free device memory by nothing; a capturing stream is refused."""
    import torch
    assert torch.cuda.is_available(), "this test needs a GPU"
    import bench
    from ilqr_amd import _lib
    from ilqr_amd.batch import BatchSolver, _ptr
    for d in range(min(torch.cuda.device_count(), 2)):       # the array's address is taken per device
        dev = torch.device("cuda", d)
        with torch.cuda.device(dev):
            s = BatchSolver(0, [2.0, 2.0], 40, 50, 20, 1e-3, device=dev)
            B = 1024
            x0, u0 = bench.car_inputs(B, 5)
            x0s = torch.as_tensor(x0, device=dev).t().contiguous()
            us = torch.as_tensor(u0, device=dev).permute(1, 2, 0).contiguous()
            x = torch.empty((40, 5, B), dtype=torch.float64, device=dev)
            out = s.alloc_outputs(B)
            oo = _lib.Outputs(*[_ptr(out.get(k)) for k, _ in _lib.Outputs._fields_])
            pr = s._problem(B, B)
            u = us.clone()
            u2 = us.clone()
            s._check(s.lib.ilqr_solve_batch_static(C.byref(pr), _ptr(x0s), _ptr(u), _ptr(x), C.byref(oo), s._stream()))
            torch.cuda.synchronize(dev)
            free0, _ = torch.cuda.mem_get_info(dev)
            s._check(s.lib.ilqr_solve_batch_static(C.byref(pr), _ptr(x0s), _ptr(u2), _ptr(x), C.byref(oo), s._stream()))
            torch.cuda.synchronize(dev)
            free1, _ = torch.cuda.mem_get_info(dev)
            assert free1 == free0, (d, free0, free1)
            assert torch.equal(u, u2)
            # under stream capture the call is refused before it enqueues anything
            side = torch.cuda.Stream(device=dev)
            g = torch.cuda.CUDAGraph()
            u3 = us.clone()
            with torch.cuda.stream(side):
                g.capture_begin()
                try:
                    rc = s.lib.ilqr_solve_batch_static(C.byref(pr), _ptr(x0s), _ptr(u3), _ptr(x), C.byref(oo),
                                                       C.c_void_p(side.cuda_stream))
                    msg = s.lib.ilqr_last_error()
                finally:
                    g.capture_end()
            assert rc == -1 and b"captured" in msg, (rc, msg)
