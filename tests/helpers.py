"""Shared comparison logic for the parity tests."""
import numpy as np

RTOL = 1e-8        # parity bar: final cost and x/u trajectories within 1e-8 relative
RTOL_SENSITIVE = 1e-6   # for solves the reference itself moves by > SENS_TIGHT under a 1e-14 input perturbation
SENS_TIGHT = 1e-10
SENS_FACTOR = 100.0     # ... and by more than 1e-8: the bar follows the reference's own measured sensitivity (`sens` is
                        # the relative change of its result under ONE 1e-14 perturbation of the inputs)
RTOL_UNCONVERGED = 1e-4   # solves that ran out of iterations (exit status 2): rounding noise accumulates over
                          # up to max_iter sweeps of a non-converging iteration; decisions must still be identical
RTOL_MEANINGLESS = 1e-2   # a bar above this compares nothing: such solves are held to their decisions only


def tolerance(sens, status):
    tol = np.where(sens <= SENS_TIGHT, RTOL, np.maximum(RTOL_SENSITIVE, SENS_FACTOR * sens))
    return np.where(status == 2, np.maximum(tol, RTOL_UNCONVERGED), tol)


def rel_traj(a, b):
    """max |a-b| over each trajectory / max |b| over that trajectory."""
    B = a.shape[0]
    num = np.abs(a - b).reshape(B, -1).max(1)
    den = np.maximum(np.abs(b).reshape(B, -1).max(1), 1e-300)
    return num / den


class Compared(int):
    """Number of solves compared one by one, with `within_1e8` = how many of them have x, u and J within 1e-8
    (relative) of the reference's, whatever bar they were held to, and `stable` = how many solves of the fixture the
    reference itself reproduces to 1e-8 under a 1e-14 perturbation of its inputs."""
    within_1e8 = 0
    stable = 0


def compare_to_golden(g, x, u, J, iters, status, ls_evals, accepted, regu_rounds, J0=None, label=""):
    """g: golden npz; the rest: arrays from the implementation under test.

    * exit status must agree where the reference raised LinAlgError on a non-finite initial trajectory
    * trajectories whose decisions the reference itself reproduces under a 1e-14 input perturbation
      (g['dstable']): identical iterations, line-search evaluations, acceptance
      flags, regularisation rounds and exit status; x, u and final J within RTOL
      (1e-8) where that perturbation moved the reference's own result by at most
      SENS_TIGHT, within max(1e-6, 100 x that movement) for the remaining ones (the 12-state plants amplify a
      1e-14 perturbation to 1e-5 over 40 knots without a single decision changing), and within
      RTOL_UNCONVERGED (1e-4) for solves that stopped at max_iter without converging.
    Solves whose result the perturbation moved by more than 1e-4 are not compared.  Returns the number compared.
    """
    B = len(g["iters"])
    st_ref = g["status"]
    lin = st_ref == 3
    # a LinAlgError on a non-finite initial trajectory must be reproduced; one reached by regu overflowing
    # after thousands of doublings belongs to the ill-conditioned solves and is not compared
    lin0 = lin & (g["regu_rounds"] == 0)
    assert np.array_equal(status[lin0], st_ref[lin0]), f"{label}: LinAlgError cases differ"
    # comparable: the reference reproduced its decisions under the perturbation and its result moved by less than
    # 1e-4 (beyond that the bar on x, u, J would exceed 1e-2 and one perturbation is no evidence that the decisions
    # of up to max_iter sweeps are pinned)
    ok = (g["dstable"] == 1) & ~lin & (SENS_FACTOR * g["sens"] <= RTOL_MEANINGLESS)
    if J0 is not None and ok.any():
        assert np.allclose(J0[ok], g["J0"][ok], rtol=RTOL, atol=0), f"{label}: initial cost differs"
    T = min(ls_evals.shape[1], g["ls_evals"].shape[1])
    bad = []
    for b in np.where(ok)[0]:
        same = (iters[b] == g["iters"][b] and status[b] == st_ref[b]
                and np.array_equal(ls_evals[b, :T], g["ls_evals"][b, :T])
                and np.array_equal(accepted[b, :T], g["accepted"][b, :T])
                and regu_rounds[b] == g["regu_rounds"][b])
        if not same:
            bad.append((int(b), "decisions", int(iters[b]), int(g["iters"][b]), int(regu_rounds[b]),
                        int(g["regu_rounds"][b])))
    # `dstable` is one sample (one perturbation): a solve that sits on a threshold (expected reduction next to 1,
    # Armijo test next to equality) can pass it and still be decided by rounding.  One such solve per 32 compared
    # may differ, and only if the perturbation moved its result measurably (sens above SENS_TIGHT).
    sensitive = all(g["sens"][b[0]] > SENS_TIGHT for b in bad)
    assert len(bad) <= max(1, int(ok.sum()) // 32) and sensitive, \
        f"{label}: decision mismatch on stable trajectories: {bad[:5]}"
    for b in bad:
        ok[b[0]] = False
    if ok.any():
        rx = rel_traj(x[ok], g["x"][ok])
        ru = rel_traj(u[ok], g["u"][ok])
        Jr = g["J_last"][ok]
        # the solver's J can legitimately be NaN (J = Jn after an abandoned search, iLQRController.py:79)
        assert np.array_equal(np.isnan(J[ok]), np.isnan(Jr)), f"{label}: NaN pattern of J differs"
        fin = ~np.isnan(Jr)
        rJ = np.abs(J[ok][fin] - Jr[fin]) / np.maximum(np.abs(Jr[fin]), 1e-300)
        rJ = np.append(rJ, 0.0)
        tol = tolerance(g["sens"][ok], st_ref[ok])
        assert (rx <= tol).all(), f"{label}: x differs by {rx.max():.3e} (worst ratio to tol {np.max(rx / tol):.2f})"
        assert (ru <= tol).all(), f"{label}: u differs by {ru.max():.3e}"
        assert (rJ[:-1] <= tol[fin]).all(), f"{label}: J differs by {rJ.max():.3e}"
        rJall = np.zeros(len(rx))
        rJall[fin] = rJ[:-1]
        tight = int(((rx <= RTOL) & (ru <= RTOL) & (rJall <= RTOL)).sum())
    else:
        tight = 0
    n = Compared(int(ok.sum()))
    n.within_1e8 = tight
    n.stable = int(np.sum(g["stable"] == 1)) if "stable" in getattr(g, "files", g) else -1
    print(f"[parity] {label or 'fixture'}: {int(n)} of {B} solves compared one by one, {tight} within 1e-8 on x, u and J "
          f"(the reference reproduces itself to 1e-8 on {n.stable})")
    return n


def model_key(g):
    return str(g["model"])


# Fixtures at the contract's horizons (Car N = 200, Euler-angle quadrotor N = 100, quaternion quadrotor N = 80 and
# its 50-step receding-horizon sequence): the reference does not reproduce its own decisions there under a 1e-14
# perturbation, from far starts or from near the goal (thousands of regularisation rounds per solve, |Q| up to
# 1e308).  They pin exit classes (a LinAlgError on the first backward stage must be reproduced) and how many
# solves are comparable at all; every other fixture must have at least half of its solves compared.
STATUS_CLASS_FIXTURES = dict(car_N200=0, car_circle_N200=0, quad_ea_N100=0, quad_quat_N80=0, quad_ea_hover_N100=0,
                             quad_quat_hover_N80=0, quad_quat_mpc80=0, car_nan=0)


def min_compared(name, g):
    """How many solves of fixture `name` a golden test must have compared for its pass to mean something."""
    if name in STATUS_CLASS_FIXTURES:
        return STATUS_CLASS_FIXTURES[name]
    return (len(g["iters"]) + 1) // 2


def min_within_1e8(name, g):
    """The 1e-8 bar as a number per fixture: of the solves the reference itself reproduces to 1e-8 when its inputs
    are changed in the 14th digit (`stable`), all but one in sixteen (at least one: `stable` is a single sample) must
    come out within 1e-8 on x, u and J.  Census of `stable` (make_golden.py prints it): car fixtures 6/8 .. 256 of
    256 per fixture (car_box256: 210 of 256), acrobot 14..32 of 16..32, quadrotors at N = 30: 4..11 of 16, at N = 40 from hover: 0 of 16
    (the reference's unsymmetrised V_xx recursion amplifies rounding by 2.3 per stage for those plants), at N = 16
    near hover: 32 of 32."""
    if name in STATUS_CLASS_FIXTURES:
        return 0
    stable = int(np.sum(g["stable"] == 1))
    return stable - max(1, stable // 16)


def cost_horizon(g):
    """Knot points the fixture's plant sums its trajectory cost over (0: the solver's horizon).  Fixtures of the acrobot
    and the quadrotors record the horizon their plant was built with (`model_N`)."""
    if "model_N" in getattr(g, "files", g):
        return 0 if int(g["model_N"]) == int(g["N"]) else int(g["model_N"])
    return 0
