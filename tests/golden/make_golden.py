#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by running the REFERENCE solver.

Run in the build container only (needs the reference tree, default
/root/reference, on disk; it is imported, never copied):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--only car_readme,...]

What is the reference's own code here and what is not:
  * `control.iLQRController.iLQR` and `control.models.model.Model` are imported
    unmodified from the reference tree; every solver-level decision in the
    fixtures (iterations, line-search evaluations, acceptance, regularisation
    rounds, exit path, the LAPACK eigenvalue test) is theirs.
  * The plant model files cannot be imported (pydrake / CasADi are not
    installed), so the plants are the NumPy restatements in
    oracle/models_numpy.py, subclassing the imported `Model`.
Instrumentation wraps bound methods and `np.linalg.eigvals` from the outside;
no reference file is edited.

Each fixture is an .npz of plain arrays: inputs (x0, u_init, parameter block),
outputs (x, u, J0, per-iteration J / deltaJ / ls_evals / accepted, iters,
status, regu_rounds, max_abs_q, lam_margin) and two flags from repeating the same
solve in the reference with x0 and u_init perturbed in the 14th digit: `dstable`,
the decision counts (iterations, line-search evaluations and acceptances,
regularisation rounds, exit) came out identical, and `stable`, in addition
trajectories and cost moved by at most 1e-8 (`sens` holds the observed relative
change).  Solves with dstable == 0 are not reproducible by the reference itself and
are compared on exit status only; for the others the bar on x, u and J follows
`sens` (tests/helpers.py).
"""
from __future__ import annotations

import argparse
import os
import sys
import warnings
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("ILQR_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
TRACE = 64
OUT = os.path.join(ROOT, "tests", "golden")

_M = None


def models():
    global _M
    if _M is None:
        from control.models.model import Model
        from oracle.models_numpy import make_models
        _M = make_models(Model)
    return _M


_USER_FN = {}


def make_user_plant(spec):
    """A plant written by the user (tests/user_models.py) as a subclass of the imported reference `Model`:
    its own Python dynamics and costs, derivatives from SymPy lambdify of the same expressions
    (the car.py:30-52 recipe with SymPy in place of pydrake.symbolic)."""
    import sympy as sp
    from control.models.model import Model
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import user_models
    um = getattr(user_models, spec["cls"])(*spec.get("args", ()))
    key = (spec["cls"], tuple(um.p))
    if key not in _USER_FN:
        xs, us, ps, f, l, lf = um.trace()[:6]
        sub = dict(zip(ps, um.p))
        f, l, lf = f.subs(sub), l.subs(sub), lf.subs(sub)
        lx = sp.Matrix([l]).jacobian(xs); lu = sp.Matrix([l]).jacobian(us); lfx = sp.Matrix([lf]).jacobian(xs)
        ex = dict(l_x=lx, l_u=lu, l_xx=lx.jacobian(xs), l_ux=lu.jacobian(xs), l_uu=lu.jacobian(us),
                  f_x=f.jacobian(xs), f_u=f.jacobian(us), lf_x=lfx, lf_xx=lfx.jacobian(xs))
        _USER_FN[key] = {k: sp.lambdify(list(xs) + (list(us) if not k.startswith("lf") else []), v, "numpy")
                         for k, v in ex.items()}
    fn = _USER_FN[key]
    n, m = um.nx, um.nu

    class Plant(Model):
        def __init__(self):
            self.nx, self.nu, self.ndx = n, m, um.ndx
            self.p = um.p

        def discrete_dynamics(self, x, u):
            return um.python_dynamics(x, u)

        def CalcDifferentialState(self, state, ref_state):
            if um._user_diff is None:
                return Model.CalcDifferentialState(self, state, ref_state)
            return np.asarray(um._user_diff(um, np.asarray(state, dtype=np.float64), np.asarray(ref_state, dtype=np.float64)),
                              dtype=np.float64)

        def _cost_stage(self, x, u):
            return um.python_cost_stage(x, u)

        def _cost_final(self, x):
            return um.python_cost_final(x)

        def stage(self, x, u):
            a = (*x, *u)
            g = lambda k, shp: np.asarray(fn[k](*a), dtype=np.float64).reshape(shp)
            return (g("l_x", n), g("l_u", m), g("l_xx", (n, n)), g("l_ux", (m, n)), g("l_uu", (m, m)),
                    g("f_x", (n, n)), g("f_u", (n, m)))

        def final(self, x):
            return (np.asarray(fn["lf_x"](*x), dtype=np.float64).reshape(n),
                    np.asarray(fn["lf_xx"](*x), dtype=np.float64).reshape(n, n))

    if um._user_own:
        # the plant's own stage / final / Q_terms, run as written on NumPy arrays by the reference solver
        # (the overrides of control/models/Quadrotor_Quat.py:248-275)
        f64 = lambda t: tuple(np.asarray(a, dtype=np.float64) for a in t)
        Plant.stage = lambda self, x, u: f64(um._user_own["stage"](um, np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)))
        Plant.final = lambda self, x: f64(um._user_own["final"](um, np.asarray(x, dtype=np.float64)))
        Plant.Q_terms = lambda self, x, *a: f64(um._user_own["Q_terms"](um, np.asarray(x, dtype=np.float64), *a))
    return Plant()


def make_plant(spec):
    if spec["model"] == "user":
        return make_user_plant(spec)
    M = models()
    kind = spec["model"]
    if kind == "car":
        return M["CarModel"](spec["r"], spec["v_target"])
    cls = dict(acrobot="Acrobot2DModel", quad_ea="QuadrotorEAModel", quad_quat="QuadrotorQuatModel")[kind]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        p = M[cls](Q=spec["Q"], Qf=spec["Qf"], R=spec["R"], dt=spec["dt"], N=spec["N"],
                   integration=spec.get("integration", "RK4"))
    p.set_references(spec["xref"], spec["uref"])
    return p


def par_block(spec):
    from oracle import c_oracle as co
    if spec["model"] == "user":
        return np.asarray(make_user_plant(spec).p, dtype=np.float64)
    if spec["model"] == "car":
        return co.pack_car(spec["r"], spec["v_target"])
    return co.pack_quadratic(spec["dt"], spec["Q"], spec["Qf"], spec["R"], spec["xref"], spec["uref"],
                             euler=spec.get("integration", "RK4") != "RK4")


def run_reference(plant, sp, x0, u_init):
    """One instrumented run of the reference's run_ilqr.  Returns a dict of plain values."""
    from control.iLQRController import iLQR
    ctl = iLQR(plant, sp["N"], sp["max_iter"], sp["max_ls"], sp["d_tol"])
    ev = []
    eig_stats = dict(n=0, neg=0, max_abs_q=0.0, margin=np.inf, cplx=0)
    orig_eig = np.linalg.eigvals

    def eig(Q):
        w = orig_eig(Q)
        eig_stats["n"] += 1
        neg = bool(np.any(w < 0))
        eig_stats["neg"] += neg
        a = float(np.abs(Q).max())
        eig_stats["max_abs_q"] = max(eig_stats["max_abs_q"], a)
        if a > 0:
            eig_stats["margin"] = min(eig_stats["margin"], float(np.abs(w.real).min()) / a)
        eig_stats["cplx"] += int(np.iscomplexobj(w))
        return w

    bp, fp, ct = ctl._backward_pass, ctl._forward_pass, plant.cost_trj

    def wbp(*a):
        r = bp(*a)
        ev.append(("bp", float(np.squeeze(r[2])), float(np.max(np.linalg.norm(r[0], axis=1)))))
        return r

    def wfp(*a):
        ev.append(("fp", float(a[4])))
        return fp(*a)

    def wct(*a):
        c = ct(*a)
        ev.append(("c", float(np.squeeze(c))))
        return c

    ctl._backward_pass, ctl._forward_pass, plant.cost_trj = wbp, wfp, wct
    np.linalg.eigvals = eig
    u = np.array(u_init, dtype=np.float64, copy=True)
    status = None
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                x, u2 = ctl.run_ilqr(np.array(x0, dtype=np.float64), u)
        assert u2 is u
        x = np.asarray(x, dtype=np.float64)
    except np.linalg.LinAlgError:
        status = 3
        x = np.full((sp["N"], plant.nx), np.nan)
    finally:
        np.linalg.eigvals = orig_eig
        del plant.cost_trj
    # decode the event stream
    J0 = ev[0][1]
    J = np.zeros(TRACE); dJ = np.zeros(TRACE); ls = np.zeros(TRACE, np.int32); acc = np.zeros(TRACE, np.uint8)
    iters = 0
    Jcur = J0
    i = 1
    last_dJ = None
    dmax = np.inf
    while i < len(ev):
        assert ev[i][0] == "bp", ev[i]
        iters += 1
        last_dJ = ev[i][1]
        dmax = ev[i][2]
        if iters <= TRACE:
            dJ[iters - 1] = last_dJ
        i += 1
        n_ls = 0
        Jn = None
        alpha = None
        while i < len(ev) and ev[i][0] == "fp":
            alpha = ev[i][1]
            Jn = ev[i + 1][1]
            n_ls += 1
            i += 2
        if n_ls and iters <= TRACE:
            ls[iters - 1] = n_ls
            J[iters - 1] = Jn
            acc[iters - 1] = (Jn - Jcur <= -1e-2 * alpha * last_dJ) and (n_ls < sp["max_ls"])
        if n_ls:
            Jcur = Jn
    if status is None:
        if iters and n_ls == 0:
            status = 1            # left through `if deltaJ < 1: break` (no forward pass after the last sweep)
        elif not (dmax > sp["d_tol"]):
            status = 0            # the while test looks at d_tol first
        else:
            status = 2
    return dict(x=x, u=u, J0=J0, J=J, deltaJ=dJ, ls_evals=ls, accepted=acc, iters=iters, status=status,
                J_last=Jcur, dmax=dmax, regu_rounds=eig_stats["neg"], n_eig=eig_stats["n"],
                max_abs_q=eig_stats["max_abs_q"], lam_margin=eig_stats["margin"], n_complex=eig_stats["cplx"])


def _one(args):
    spec, sp, x0, u0, check_stable = args
    plant = make_plant(spec)
    r = run_reference(plant, sp, x0, u0)
    stable, sens, dstable = 1, 0.0, 1
    if check_stable:
        # the same solve with every input perturbed in the 14th digit
        rng = np.random.default_rng(12345)
        up = u0 * (1.0 + 1e-14 * rng.choice([-1.0, 1.0], size=u0.shape))
        xp = x0 * (1.0 + 1e-14 * rng.choice([-1.0, 1.0], size=x0.shape))
        r2 = run_reference(make_plant(spec), sp, xp, up)
        same = (r["iters"] == r2["iters"] and r["status"] == r2["status"]
                and np.array_equal(r["ls_evals"], r2["ls_evals"]) and np.array_equal(r["accepted"], r2["accepted"])
                and r["regu_rounds"] == r2["regu_rounds"])
        dstable = int(same)
        if r["status"] != 3 and r2["status"] != 3:
            sens = float(np.abs(r["x"] - r2["x"]).max() / max(1e-300, np.abs(r["x"]).max()))
            sens = max(sens, float(np.abs(r["u"] - r2["u"]).max() / max(1e-300, np.abs(r["u"]).max())))
            sens = max(sens, abs(r["J_last"] - r2["J_last"]) / max(1e-300, abs(r["J_last"])))
            same = same and sens <= 1e-8
        stable = int(same)
    r["sens"] = sens
    r["stable"] = stable
    r["dstable"] = dstable
    return r


def run_cases(name, spec, sp, X0, U0, pool, check_stable=True, extra=None):
    B = len(X0)
    res = list(pool.map(_one, [(spec, sp, X0[b], U0[b], check_stable) for b in range(B)], chunksize=1))
    out = dict(par=par_block(spec), model=np.array(spec["model"]), N=sp["N"], max_iter=sp["max_iter"],
               max_ls=sp["max_ls"], d_tol=sp["d_tol"], x0=np.asarray(X0), u_init=np.asarray(U0))
    for k in ("x", "u", "J", "deltaJ", "ls_evals", "accepted"):
        out[k] = np.stack([r[k] for r in res])
    for k in ("J0", "J_last", "dmax", "max_abs_q", "lam_margin", "sens"):
        out[k] = np.array([r[k] for r in res], dtype=np.float64)
    for k in ("iters", "status", "regu_rounds", "n_eig", "n_complex", "stable", "dstable"):
        out[k] = np.array([r[k] for r in res], dtype=np.int32)
    if spec["model"] not in ("car", "user"):
        out["model_N"] = spec["N"]        # the horizon the PLANT was built with: its cost_trj sums over that one
    if extra:
        out.update(extra)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    st = out["status"]
    print(f"[golden] {name}: B={B} iters={np.bincount(out['iters'])[:12]}... status={np.bincount(st, minlength=4)} "
          f"stable={out['stable'].sum()}/{B} ls>1: {(out['ls_evals'] > 1).any(axis=1).sum()} "
          f"regu>0: {(out['regu_rounds'] > 0).sum()}", flush=True)
    return out


# ----------------------------------------------------------------------------
CAR = dict(model="car", r=2.0, v_target=2.0)
CAR_SP = dict(N=40, max_iter=50, max_ls=20, d_tol=1e-3)


def car_box(rng, B):
    return np.column_stack([rng.uniform(-4, 4, B), rng.uniform(-4, 4, B), rng.uniform(-np.pi, np.pi, B),
                            rng.uniform(0, 3, B), rng.uniform(-0.3, 0.3, B)])


def acrobot_spec(N=100):
    return dict(model="acrobot", Q=np.diag([1.0, 1.0, 0.1, 0.1]), Qf=100.0 * np.eye(4), R=np.array([[0.01]]),
                dt=0.05, N=N, xref=np.array([np.pi, 0, 0, 0.0]), uref=np.zeros(1))


def quad_weights():
    Q = np.diag([10.0] * 6 + [1.0] * 6)
    Qf = np.diag([500.0] * 3 + [100.0] * 3 + [20.0] * 6)
    R = np.diag([0.02] * 4)
    return Q, Qf, R


def ea_spec(N=100):
    Q, Qf, R = quad_weights()
    xref = np.zeros(12); xref[2] = 1.0
    return dict(model="quad_ea", Q=Q, Qf=Qf, R=R, dt=0.01, N=N, xref=xref, uref=np.zeros(4))


def quat_spec(N=80):
    Q, Qf, R = quad_weights()
    xref = np.zeros(13); xref[0] = 1.0; xref[6] = 1.0
    return dict(model="quad_quat", Q=Q, Qf=Qf, R=R, dt=0.01, N=N, xref=xref, uref=np.zeros(4))


def euler_zyx_to_quat(e):
    """unit quaternion [w,x,y,z] from ZYX Euler angles e = (yaw, pitch, roll)"""
    cy, sy = np.cos(e[0] / 2), np.sin(e[0] / 2)
    cp, sp_ = np.cos(e[1] / 2), np.sin(e[1] / 2)
    cr, sr = np.cos(e[2] / 2), np.sin(e[2] / 2)
    return np.array([cr * cp * cy + sr * sp_ * sy, sr * cp * cy - cr * sp_ * sy,
                     cr * sp_ * cy + sr * cp * sy, cr * cp * sy - sr * sp_ * cy])


def gen_car_readme(pool):
    x0 = np.array([[-3.0, 1.0, -0.2, 0.0, 0.0]])
    u0 = np.random.default_rng(0).standard_normal((1, 39, 2)) * 1e-4
    run_cases("car_readme", CAR, CAR_SP, x0, u0, pool)


def gen_car_mpc(pool):
    """Driving.ipynb cell-2 protocol: 50 re-solves from x[1], u carried over without shifting."""
    plant = make_plant(CAR)
    x0 = np.array([-3.0, 1.0, -0.2, 0.0, 0.0])
    u = np.random.default_rng(0).standard_normal((39, 2)) * 1e-4
    X0, U0 = [], []
    for _ in range(50):
        X0.append(x0.copy()); U0.append(u.copy())
        r = run_reference(plant, CAR_SP, x0, u)
        u = r["u"]; x0 = r["x"][1].copy()
    run_cases("car_mpc50", CAR, CAR_SP, np.array(X0), np.array(U0), pool)


def gen_car_box(pool):
    rng = np.random.default_rng(1234)
    B = 256
    run_cases("car_box256", CAR, CAR_SP, car_box(rng, B), rng.standard_normal((B, 39, 2)) * 1e-4, pool)


def gen_car_adversarial(pool):
    rng = np.random.default_rng(77)
    B = 32
    X0 = car_box(rng, B)
    # large random controls: forces backtracking and regularisation
    run_cases("car_bigu", CAR, CAR_SP, X0, rng.standard_normal((B, 39, 2)) * 2.0, pool)
    # max_ls = 1 / 2: every search is abandoned / only alpha=1 can be accepted
    run_cases("car_maxls1", CAR, dict(CAR_SP, max_ls=1), X0[:8], rng.standard_normal((8, 39, 2)) * 0.5, pool)
    run_cases("car_maxls2", CAR, dict(CAR_SP, max_ls=2), X0[:8], rng.standard_normal((8, 39, 2)) * 2.0, pool)
    # max_iter exhaustion
    run_cases("car_maxiter2", CAR, dict(CAR_SP, max_iter=2), X0[:8], rng.standard_normal((8, 39, 2)) * 1e-4, pool)
    # loose d_tol so the loop ends through the d_tol test
    run_cases("car_dtol", CAR, dict(CAR_SP, d_tol=50.0), X0[:16], rng.standard_normal((16, 39, 2)) * 1e-4, pool)
    # huge controls: overflow -> inf/NaN initial rollout -> LinAlgError in eigvals
    u = rng.standard_normal((4, 39, 2)) * 1e155
    run_cases("car_nan", CAR, CAR_SP, X0[:4], u, pool, check_stable=False)
    # short and long horizons
    run_cases("car_N2", CAR, dict(CAR_SP, N=2), X0[:8], rng.standard_normal((8, 1, 2)) * 0.1, pool)
    run_cases("car_N200", CAR, dict(CAR_SP, N=200), X0[:8], rng.standard_normal((8, 199, 2)) * 1e-4, pool)


def gen_acrobot(pool):
    rng = np.random.default_rng(2)
    B = 32
    sp = dict(N=100, max_iter=20, max_ls=10, d_tol=1e-3)
    X0 = np.column_stack([rng.uniform(-np.pi, np.pi, B), rng.uniform(-np.pi, np.pi, B),
                          rng.uniform(-1, 1, B), rng.uniform(-1, 1, B)])
    run_cases("acrobot_N100", acrobot_spec(100), sp, X0, rng.standard_normal((B, 99, 1)), pool)
    run_cases("acrobot_N40", acrobot_spec(40), dict(sp, N=40), X0[:16], rng.standard_normal((16, 39, 1)), pool)


def gen_quad_ea(pool):
    rng = np.random.default_rng(3)
    B = 16
    sp = dict(N=100, max_iter=20, max_ls=10, d_tol=5e-4)
    X0 = np.zeros((B, 12)); X0[:, 2] = 0.8; X0[:, 3:6] = 0.5 * rng.standard_normal((B, 3))
    run_cases("quad_ea_N100", ea_spec(100), sp, X0, 2.0 * rng.standard_normal((B, 99, 4)), pool)
    run_cases("quad_ea_N30", ea_spec(30), dict(sp, N=30), X0, 2.0 * rng.standard_normal((B, 29, 4)), pool)


def gen_quad_quat(pool):
    rng = np.random.default_rng(4)
    B = 16
    sp = dict(N=80, max_iter=20, max_ls=10, d_tol=5e-4)
    X0 = np.zeros((B, 13)); X0[:, 6] = 0.8
    for b in range(B):
        X0[b, :4] = euler_zyx_to_quat(0.5 * rng.standard_normal(3))
    run_cases("quad_quat_N80", quat_spec(80), sp, X0, 2.0 * rng.standard_normal((B, 79, 4)), pool)
    run_cases("quad_quat_N30", quat_spec(30), dict(sp, N=30), X0, 2.0 * rng.standard_normal((B, 29, 4)), pool)
    # receding horizon (QuadrotorControllers.py:173-191): solve, step the plant with u[0], shift u, re-solve
    spec = quat_spec(30)
    spm = dict(sp, N=30)
    plant = make_plant(spec)
    X0s, U0s = [], []
    for b in range(2):
        x0 = X0[b].copy(); u = 2.0 * rng.standard_normal((29, 4))
        for step in range(6):
            X0s.append(x0.copy()); U0s.append(u.copy())
            r = run_reference(plant, spm, x0, u)
            u = r["u"]
            x0 = np.asarray(plant.discrete_dynamics(x0, u[0])).ravel()
            u[:-1] = u[1:]
    run_cases("quad_quat_mpc", spec, spm, np.array(X0s), np.array(U0s), pool,
              extra=dict(mpc_steps=6, mpc_traj=2))


def gen_baseline_horizons(pool):
    """The contract's configurations 3, 4 and 5 at their stated horizons from starts the reference solves
    solves as reproducibly as it can: quadrotors near hover with controls around the hover thrust (m g / 4 = 1.9), the
    car near the circle it tracks.  At the stated horizons (N = 100, 80, 200) the reference does not reproduce its own
    decisions under a 1e-14 perturbation even from these starts (thousands of regularisation rounds per solve, |Q| up
    to 1e308): those fixtures pin exit classes; the same starts at N = 40 / 40 / 60 are reproducible and pin decisions
    and trajectories."""
    rng = np.random.default_rng(33)
    B = 16
    sp = dict(N=100, max_iter=20, max_ls=10, d_tol=5e-4)
    X0 = np.zeros((B, 12)); X0[:, 2] = 0.8; X0[:, 3:6] = 0.2 * rng.standard_normal((B, 3))
    X0[:, :2] = 0.2 * rng.standard_normal((B, 2))
    run_cases("quad_ea_hover_N100", ea_spec(100), sp, X0, 1.9 + 0.5 * rng.standard_normal((B, 99, 4)), pool)
    run_cases("quad_ea_hover_N40", ea_spec(40), dict(sp, N=40), X0, 1.9 + 0.5 * rng.standard_normal((B, 39, 4)), pool)
    spq = dict(sp, N=80)
    X0 = np.zeros((B, 13)); X0[:, 6] = 0.8; X0[:, 4:6] = 0.2 * rng.standard_normal((B, 2))
    for b in range(B):
        X0[b, :4] = euler_zyx_to_quat(0.2 * rng.standard_normal(3))
    U0 = 1.9 + 0.5 * rng.standard_normal((B, 79, 4))
    run_cases("quad_quat_hover_N80", quat_spec(80), spq, X0, U0, pool)
    run_cases("quad_quat_hover_N40", quat_spec(40), dict(sp, N=40), X0, U0[:, :39], pool)
    # configuration 4: N = 80, 50 steps of receding horizon (QuadrotorControllers.py:173-191): solve, apply u[0]
    # through the plant's own discrete_dynamics, shift u, re-solve
    spec = quat_spec(80)
    plant = make_plant(spec)
    X0s, U0s = [], []
    x0 = X0[0].copy(); u = U0[0].copy()
    for step in range(50):
        X0s.append(x0.copy()); U0s.append(u.copy())
        r = run_reference(plant, spq, x0, u)
        u = r["u"]
        x0 = np.asarray(plant.discrete_dynamics(x0, u[0])).ravel()
        u[:-1] = u[1:]
    run_cases("quad_quat_mpc80", spec, spq, np.array(X0s), np.array(U0s), pool, extra=dict(mpc_steps=50, mpc_traj=1))
    # configuration 5: Car N = 200 from starts near the circle, heading along it
    B = 16
    a = rng.uniform(-np.pi, np.pi, B); rad = 2.0 + 0.2 * rng.standard_normal(B)
    X0 = np.column_stack([rad * np.cos(a), rad * np.sin(a), a + np.pi / 2 + 0.1 * rng.standard_normal(B),
                          2.0 + 0.2 * rng.standard_normal(B), 0.05 * rng.standard_normal(B)])
    run_cases("car_circle_N200", CAR, dict(CAR_SP, N=200), X0, rng.standard_normal((B, 199, 2)) * 1e-4, pool)
    run_cases("car_circle_N60", CAR, dict(CAR_SP, N=60), X0, rng.standard_normal((B, 59, 2)) * 1e-4, pool)


def gen_variants(pool):
    """What the plants' constructors accept beyond the notebooks' settings: `integration` other than "RK4" (one Euler
    step, acrobot.py:83-86), weight matrices that are neither diagonal nor symmetric (Quadrotor_Quat.py:152 takes any
    Q), and a plant built with a shorter horizon than the solver runs (cost_trj sums over the plant's, acrobot.py:139).
    Starts near the goal / near hover, so that the solves are reproducible."""
    rng = np.random.default_rng(44)
    B = 16
    sp = dict(N=30, max_iter=20, max_ls=10, d_tol=5e-4)
    # acrobot, Euler, from near the upright goal
    X0 = np.array([np.pi, 0, 0, 0]) + 0.3 * rng.standard_normal((B, 4))
    spa = dict(N=40, max_iter=20, max_ls=10, d_tol=1e-3)
    run_cases("acrobot_euler_N40", dict(acrobot_spec(40), integration="Euler"), spa, X0,
              0.5 * rng.standard_normal((B, 39, 1)), pool)
    # plant horizon 25, solver horizon 40
    run_cases("acrobot_modelN25_N40", acrobot_spec(25), spa, X0, 0.5 * rng.standard_normal((B, 39, 1)), pool)
    # quadrotors near hover
    Xe = np.zeros((B, 12)); Xe[:, 2] = 0.8; Xe[:, 3:6] = 0.2 * rng.standard_normal((B, 3))
    Xe[:, :2] = 0.2 * rng.standard_normal((B, 2))
    Ue = 1.9 + 0.5 * rng.standard_normal((B, 29, 4))
    run_cases("quad_ea_euler_N30", dict(ea_spec(30), integration="Euler"), sp, Xe, Ue, pool)
    Xq = np.zeros((B, 13)); Xq[:, 6] = 0.8; Xq[:, 4:6] = 0.2 * rng.standard_normal((B, 2))
    for b in range(B):
        Xq[b, :4] = euler_zyx_to_quat(0.2 * rng.standard_normal(3))
    run_cases("quad_quat_euler_N30", dict(quat_spec(30), integration="Euler"), sp, Xq, Ue, pool)
    # full weights: the notebook's diagonals plus couplings that are not symmetric
    spec = quat_spec(30)
    C = rng.standard_normal((12, 12))
    spec["Q"] = spec["Q"] + 0.3 * C
    spec["Qf"] = spec["Qf"] + 3.0 * rng.standard_normal((12, 12))
    spec["R"] = spec["R"] + 0.004 * rng.standard_normal((4, 4))
    run_cases("quad_quat_fullw_N30", spec, sp, Xq, Ue, pool)


def gen_near_goal(pool):
    """Both quadrotors from starts a few centimetres and hundredths of a radian from hover, controls within 0.05 of
    the hover thrust, at a horizon of 16 knots.  The reference's backward pass never symmetrises V_xx, and for these
    plants (dt = 0.01, R = 0.02) the antisymmetric part of V_xx grows by a factor of about 2.3 per stage: rounding noise
    of 2e-16 at the last stage is 1e-4 of V_xx forty stages earlier, which is why not one of the N = 40 solves from hover
    is reproduced by the reference to 1e-8 under a 1e-14 perturbation of its inputs (`stable` 0 of 16), however easy the
    solve.  Sixteen knots keep that growth below 1e-9, so these solves pin x, u and J of the 12-state kernels at the
    1e-8 bar itself."""
    rng = np.random.default_rng(55)
    B = 32
    sp = dict(N=16, max_iter=20, max_ls=10, d_tol=5e-4)
    Xe = np.zeros((B, 12)); Xe[:, 2] = 1.0 + 0.05 * rng.standard_normal(B); Xe[:, :2] = 0.05 * rng.standard_normal((B, 2))
    Xe[:, 3:6] = 0.02 * rng.standard_normal((B, 3)); Xe[:, 6:9] = 0.02 * rng.standard_normal((B, 3))
    Ue = 1.9006875 + 0.05 * rng.standard_normal((B, 15, 4))
    run_cases("quad_ea_near_N16", ea_spec(16), sp, Xe, Ue, pool)
    Xq = np.zeros((B, 13)); Xq[:, 6] = 1.0 + 0.05 * rng.standard_normal(B); Xq[:, 4:6] = 0.05 * rng.standard_normal((B, 2))
    Xq[:, 10:13] = 0.02 * rng.standard_normal((B, 3))
    for b in range(B):
        Xq[b, :4] = euler_zyx_to_quat(0.02 * rng.standard_normal(3))
    run_cases("quad_quat_near_N16", quat_spec(16), sp, Xq, Ue, pool)


def gen_units(pool):
    """Pure functions of the reference solver on random blocks (iLQRController.py:125-147, model.py:45-53)."""
    from control.iLQRController import iLQR
    from control.models.model import Model
    rng = np.random.default_rng(99)

    class Dummy(Model):
        def __init__(self, n, m):
            self.nx, self.nu, self.ndx = n, m, n

    out = {}
    for tag, (n, m) in dict(car=(5, 2), acrobot=(4, 1), quad=(12, 4)).items():
        ctl = iLQR(Dummy(n, m), 10, 1)
        T = 16
        keys = ["lx", "lu", "lxx", "lux", "luu", "fx", "fu", "Vx", "Vxx", "Qx", "Qu", "Qxx", "Qux", "Quu",
                "d", "K", "Vx2", "Vxx2", "ecr", "fullH"]
        acc = {k: [] for k in keys}
        for _ in range(T):
            A = rng.standard_normal((n, n)); lxx = A @ A.T
            Bm = rng.standard_normal((m, m)); luu = Bm @ Bm.T + np.eye(m)
            Cm = rng.standard_normal((n, n)); Vxx = Cm @ Cm.T
            lx, lu, Vx = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(n)
            lux = rng.standard_normal((m, n)); fx = np.eye(n) + 0.1 * rng.standard_normal((n, n))
            fu = rng.standard_normal((n, m))
            Qx, Qu, Qxx, Qux, Quu = ctl.model.Q_terms(None, lx, lu, lxx, lux, luu, fx, fu, Vx, Vxx)
            d, K = ctl._gains(Quu, Qu, Qux)
            Vx2, Vxx2 = ctl._V_terms(Qx, Qu, Qxx, Qux, Quu, K, d)
            ecr = ctl._expected_cost_reduction(Qu, Quu, d)
            fullH = ctl._full_hessian(Qxx, Qux, Quu)
            for k, v in zip(keys, [lx, lu, lxx, lux, luu, fx, fu, Vx, Vxx, Qx, Qu, Qxx, Qux, Quu, d, K, Vx2, Vxx2,
                                   ecr, fullH]):
                acc[k].append(np.asarray(v, dtype=np.float64))
        for k in keys:
            out[f"{tag}_{k}"] = np.stack(acc[k])
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "units.npz"), **out)
    print("[golden] units", flush=True)


def gen_user_rover(pool):
    """A plant defined by the user (tests/user_models.py Rover): random starts around a soft obstacle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from user_models import ROVER_SP, rover_inputs
    x0, u0 = rover_inputs(48, 2024)
    run_cases("user_rover", dict(model="user", cls="Rover"), ROVER_SP, x0, u0, pool)


def gen_user_tug(pool):
    """A plant defined by the user with nx + nu = 9 (tests/user_models.py Tug): the reference solves it as written,
    the library as a 12-state, 4-control plant."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from user_models import TUG_SP, tug_inputs
    x0, u0 = tug_inputs(32, 77)
    run_cases("user_tug", dict(model="user", cls="Tug"), TUG_SP, x0, u0, pool)


def gen_user_compass(pool):
    """A plant defined by the user that overrides CalcDifferentialState (tests/user_models.py Compass)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from user_models import COMPASS_SP, compass_inputs
    x0, u0 = compass_inputs(32, 11)
    run_cases("user_compass", dict(model="user", cls="Compass"), COMPASS_SP, x0, u0, pool)


def gen_user_skiff(pool):
    """A plant defined by the user on an error state of another dimension than its state, with its own
    CalcDifferentialState, stage, final and Q_terms (tests/user_models.py Skiff)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from user_models import SKIFF_SP, skiff_inputs
    x0, u0 = skiff_inputs(32, 21)
    run_cases("user_skiff", dict(model="user", cls="Skiff"), SKIFF_SP, x0, u0, pool)


def gen_quat_utils(pool):
    """Outputs of the reference's NumPy quaternion helpers (maths/quaternions.py) on random quaternions."""
    from maths import quaternions as rq
    rng = np.random.default_rng(31)
    q = rng.standard_normal((32, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    phi = rng.standard_normal((32, 3))
    out = dict(q=q, phi=phi,
               hat=np.stack([rq.hat(v) for v in q]), conj=np.stack([rq.GetConjugate(v) for v in q]),
               left=np.stack([rq.GetLeftMatrix(v) for v in q]), G=np.stack([rq.GetAttitudeJacobian(v) for v in q]),
               p2q=np.stack([rq.ParamToQuaternion(v) for v in phi]),
               q2p=np.stack([rq.QuaternionToParam(v.copy()) for v in q]),
               rot=np.stack([rq.QuaternionToRotation(v) for v in q]))
    np.savez_compressed(os.path.join(OUT, "quat_utils.npz"), **out)
    print("[golden] quat_utils: 32 quaternions", flush=True)


GENS = dict(quat_utils=gen_quat_utils, user_rover=gen_user_rover, user_tug=gen_user_tug, user_compass=gen_user_compass, user_skiff=gen_user_skiff, car_readme=gen_car_readme, car_mpc=gen_car_mpc, car_box=gen_car_box, car_adv=gen_car_adversarial,
            acrobot=gen_acrobot, quad_ea=gen_quad_ea, quad_quat=gen_quad_quat, baseline=gen_baseline_horizons, variants=gen_variants, near_goal=gen_near_goal, units=gen_units)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    which = [w for w in a.only.split(",") if w] or list(GENS)
    with ProcessPoolExecutor(max_workers=a.jobs) as pool:
        for w in which:
            GENS[w](pool)


if __name__ == "__main__":
    main()
