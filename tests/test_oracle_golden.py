"""The CPU checker (oracle/ilqr_oracle.c) against the golden vectors produced by the
reference's own iLQR class (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from helpers import compare_to_golden, cost_horizon, min_compared, min_within_1e8, model_key


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_golden(oracle, name):
    g = load_golden(name)
    r = oracle.solve_batch(model_key(g), g["par"], int(g["N"]), int(g["max_iter"]), int(g["max_ls"]),
                           float(g["d_tol"]), g["x0"], g["u_init"], cost_N=cost_horizon(g))
    i = r["info"]
    n = compare_to_golden(g, r["x"], r["u"], i["J_last"], i["iters"], i["status"], i["ls_evals"], i["accepted"],
                          i["regu_rounds"], J0=i["J0"], label=name)
    assert n >= min_compared(name, g), (name, n)
    assert n.within_1e8 >= min_within_1e8(name, g), (name, n.within_1e8, n.stable)
    if name in ("car_readme", "car_mpc50", "car_dtol", "car_maxiter2", "car_maxls1", "car_N2", "acrobot_N40",
                "car_circle_N60", "quad_ea_N30", "quad_quat_N30"):
        assert n == len(g["iters"])      # every trajectory of these fixtures is compared


def test_readme_known_answer(oracle):
    """README / Driving.ipynb cell-2 configuration: J 214.0437 -> 24.8915 in 5 sweeps, ls evals [1,1,1,1,0]."""
    g = load_golden("car_readme")
    r = oracle.solve_batch("car", g["par"], 40, 50, 20, 1e-3, g["x0"], g["u_init"])
    i = r["info"][0]
    assert abs(i["J0"] - 214.0437) < 1e-4 and abs(i["J_last"] - 24.8915) < 1e-4
    assert i["iters"] == 5 and list(i["ls_evals"][:5]) == [1, 1, 1, 1, 0] and i["status"] == 1


def test_fixture_edge_cases_present():
    """The fixtures exercise every exit path and the line-search corner cases."""
    st = {n: load_golden(n)["status"] for n in golden_names()}
    assert (st["car_dtol"] == 0).all()           # d_tol exit
    assert (st["car_readme"] == 1).all()         # deltaJ < 1 exit
    assert (st["car_maxiter2"] == 2).all()       # max_iter exit
    assert (st["car_nan"] == 3).all()            # LinAlgError
    g = load_golden("car_maxls1")                # ls == max_ls can never be accepted
    assert g["accepted"].sum() == 0 and (g["ls_evals"][:, 0] == 1).all()
    g = load_golden("car_bigu")
    assert (g["ls_evals"] > 1).any() and (g["regu_rounds"] > 0).any()


def test_units_against_reference_pure_functions(oracle):
    """Q_terms, _gains, _V_terms, _expected_cost_reduction of the imported reference on random blocks."""
    import ctypes as C
    U = np.load(__import__("os").path.join(__import__("conftest").GOLDEN_DIR, "units.npz"))
    L = oracle.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for tag, (n, m) in dict(car=(5, 2), acrobot=(4, 1), quad=(12, 4)).items():
        T = U[f"{tag}_lx"].shape[0]
        for t in range(T):
            a = {k: np.ascontiguousarray(U[f"{tag}_{k}"][t], dtype=np.float64) for k in
                 ("lx", "lu", "lxx", "lux", "luu", "fx", "fu", "Vx", "Vxx", "Qx", "Qu", "Qxx", "Qux", "Quu", "d", "K",
                  "Vx2", "Vxx2", "ecr")}
            Qx, Qu, Qxx, Qux, Quu = np.zeros(n), np.zeros(m), np.zeros((n, n)), np.zeros((m, n)), np.zeros((m, m))
            L.ilqr_oracle_unit_qterms(n, m, *[dp(a[k]) for k in ("lx", "lu", "lxx", "lux", "luu", "fx", "fu", "Vx",
                                                                  "Vxx")], dp(Qx), dp(Qu), dp(Qxx), dp(Qux), dp(Quu))
            for got, key in ((Qx, "Qx"), (Qu, "Qu"), (Qxx, "Qxx"), (Qux, "Qux"), (Quu, "Quu")):
                assert np.allclose(got, a[key].reshape(got.shape), rtol=1e-12, atol=1e-12), (tag, key)
            d, K, Vx, Vxx, ecr = np.zeros(m), np.zeros((m, n)), np.zeros(n), np.zeros((n, n)), np.zeros(1)
            rc = L.ilqr_oracle_unit_gains_vterms(n, m, dp(a["Qx"]), dp(a["Qu"]), dp(a["Qxx"]), dp(a["Qux"]),
                                                 dp(a["Quu"]), dp(d), dp(K), dp(Vx), dp(Vxx), dp(ecr))
            assert rc == 0
            sc = max(1.0, np.abs(a["Vxx2"]).max())
            assert np.allclose(d, a["d"].ravel(), rtol=1e-9, atol=1e-11)
            assert np.allclose(K, a["K"].reshape(m, n), rtol=1e-9, atol=1e-11)
            assert np.allclose(Vx, a["Vx2"].ravel(), rtol=1e-9, atol=1e-9 * sc)
            assert np.allclose(Vxx, a["Vxx2"], rtol=1e-9, atol=1e-9 * sc)
            assert np.allclose(ecr[0], float(np.squeeze(a["ecr"])), rtol=1e-9, atol=1e-9)


def test_psd_decision_rule(oracle):
    """The stand-in for `np.any(np.linalg.eigvals(Q) < 0)`."""
    rng = np.random.default_rng(0)
    for n in (5, 7, 16):
        for _ in range(50):
            A = rng.standard_normal((n, n))
            P = A @ A.T + 0.1 * np.eye(n)                      # positive definite
            assert oracle.has_negative_eig(P) == 0
            Nn = P - (np.linalg.eigvalsh(P)[0] + 0.5) * np.eye(n)   # one clearly negative eigenvalue
            assert oracle.has_negative_eig(Nn) == 1
    Z = np.diag([1.0, 0.0, 2.0])                               # structural zero row/column: eigenvalue exactly 0
    assert oracle.has_negative_eig(Z) == 0
    assert oracle.has_negative_eig(np.array([[0.0, 1.0], [1.0, 0.0]])) == 1
    assert oracle.has_negative_eig(np.array([[1.0, np.nan], [np.nan, 1.0]])) == -1
    assert oracle.has_negative_eig(np.array([[np.inf, 0.0], [0.0, 1.0]])) == -1
