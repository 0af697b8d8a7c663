"""The C-ABI library: loads, exports every symbol include/ilqr_hip.h declares, and validates
arguments on the host side (no kernel is launched here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ilqr_hip.h")


@pytest.fixture(scope="module")
def L():
    from ilqr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _lib.lib()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ilqr_[a-z_]+)\s*\(", src)))


def test_header_functions_are_exported(L):
    names = declared_functions()
    assert len(names) >= 8
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ilqr_hip.h but not exported"
    from ilqr_amd import _lib
    assert sorted(_lib.EXPORTS) == names


def test_model_info(L):
    from ilqr_amd import _lib
    assert _lib.model_info(0) == (5, 2, 5, 2)
    # [dt, Q, Qf, R, xref, uref, euler]
    assert _lib.model_info(1) == (4, 1, 4, 1 + 16 + 16 + 1 + 4 + 1 + 1)
    assert _lib.model_info(2) == (12, 4, 12, 1 + 144 + 144 + 16 + 12 + 4 + 1)
    assert _lib.model_info(3) == (13, 4, 12, 1 + 144 + 144 + 16 + 13 + 4 + 1)
    assert L.ilqr_model_info(9, None, None, None, None) == -1
    assert b"model" in L.ilqr_last_error()


def test_struct_layout_matches_header():
    from ilqr_amd import _lib
    assert C.sizeof(_lib.Problem) == 64     # 4 int32, double, 2 int64, pointer, 4 int32
    assert C.sizeof(_lib.Outputs) == 15 * 8
    assert C.sizeof(_lib.Timing) == 6 * 8 + 6 * 4 + 2 * 4


def test_argument_validation_and_workspace(L):
    from ilqr_amd import _lib
    fake = C.c_void_p(256)   # never dereferenced: validation fails first
    def prob(**kw):
        d = dict(model_id=0, N=40, max_iter=50, max_ls=20, d_tol=1e-3, B=1024, stride=1024, params=fake,
                 trace_len=0, flags=0)
        d.update(kw)
        return _lib.Problem(**d)
    p = prob()
    S, n, m, nx = 39, 5, 2, 5
    need = 8 * 1024 * (S * m + S * m * n + 40 * nx + S * m)
    assert L.ilqr_workspace_bytes(C.byref(p)) >= need
    assert L.ilqr_workspace_bytes(C.byref(prob(N=1))) == 0
    for bad in (prob(N=1), prob(B=0), prob(stride=8), prob(model_id=7), prob(params=None), prob(cost_N=41),
                prob(cost_N=-1), prob(reserved_=7)):
        assert L.ilqr_solve_batch(C.byref(bad), fake, fake, fake, None, fake, 1 << 40, None) == -1
    assert L.ilqr_solve_batch(C.byref(p), None, fake, fake, None, fake, 1 << 40, None) == -1
    assert L.ilqr_solve_batch(C.byref(p), fake, fake, fake, None, fake, 16, None) == -2
    assert L.ilqr_solve_batch(C.byref(p), fake, fake, fake, None, None, 0, None) == -2
    assert L.ilqr_version().startswith(b"ilqr_hip")
    m = re.search(r"#define ILQR_ABI_VERSION\s+(\d+)", open(HEADER).read())
    assert m and L.ilqr_abi_version() == int(m.group(1)) == _lib.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ilqr_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.LibraryMissing):
        _lib.lib()
    with pytest.raises(_lib.LibraryMissing):
        _lib.lib(str(tmp_path / "user" / "libilqr_user.so"))


def test_no_cpu_device_path():
    from ilqr_amd.batch import BatchSolver
    with pytest.raises(RuntimeError):
        BatchSolver(0, [2.0, 2.0], 40, 50, device="cpu")


def test_flag_values_match_header_and_per_trajectory_blocks_are_validated(L):
    from ilqr_amd import _lib
    src = open(HEADER).read()
    for name, val in (("PARALLEL_LS", _lib.FLAG_PARALLEL_LS), ("LS_SEQUENTIAL", _lib.FLAG_LS_SEQUENTIAL),
                      ("TIMING", _lib.FLAG_TIMING), ("SINGLE_STREAM", _lib.FLAG_SINGLE_STREAM),
                      ("PARAMS_PER_TRAJECTORY", _lib.FLAG_PARAMS_PER_TRAJECTORY)):
        m = re.search(r"ILQR_FLAG_%s\s*=\s*(\d+)" % name, src)
        assert m and int(m.group(1)) == val, name


def test_car_parameter_block_shapes():
    import numpy as np
    from ilqr_amd import CarModel
    assert CarModel(2, 2)._param_block().shape == (2,)
    assert CarModel(np.ones(7), 2.0)._param_block().shape == (7, 2)
    assert CarModel(2.0, np.arange(3.0))._param_block().tolist() == [[2.0, 0.0], [2.0, 1.0], [2.0, 2.0]]
