"""Plants defined by the user: the reference's extension point, compiled for the GPU.

The reference lets a user solve their own problem by subclassing `Model` and
writing `discrete_dynamics`, `_cost_stage` and `_cost_final`, with derivatives
taken symbolically (README.md:25-31, control/models/car.py:12-52, :61-102).
`UserModel` keeps that workflow:

    from ilqr_amd import sym as m
    from ilqr_amd.models.user import UserModel

    class Unicycle(UserModel):
        def __init__(self, goal):
            super().__init__(nx=3, nu=2, params=goal)      # self.p: numbers that may change between solves
        def discrete_dynamics(self, x, u):
            return m.array([x[0] + 0.1 * u[0] * m.cos(x[2]), x[1] + 0.1 * u[0] * m.sin(x[2]), x[2] + 0.1 * u[1]])
        def _cost_stage(self, x, u):
            return (x[0] - self.p[0]) ** 2 + (x[1] - self.p[1]) ** 2 + 0.1 * (u[0] ** 2 + u[1] ** 2)
        def _cost_final(self, x):
            return 100 * ((x[0] - self.p[0]) ** 2 + (x[1] - self.p[1]) ** 2)

    plant = Unicycle([2.0, 1.0]); plant.build()            # hipcc, once per distinct plant; reused afterwards
    x, u = iLQR(plant, N=50, max_iter=50).run_ilqr(x0, u0)

`build()` runs the three methods once on SymPy symbols (x, u and `self.p`
become symbols), differentiates, prints straight-line C and compiles the whole
solver for that plant into its own shared library under ilqr_amd/lib/user/.
After `build()` the Model interface (discrete_dynamics, stage, final, rollout,
cost_trj, ...) evaluates on the GPU like the shipped plants; the Python methods
the user wrote stay reachable as `python_dynamics`, `python_cost_stage`,
`python_cost_final`.  Plants with nx + nu <= 8 are solved one trajectory per
lane; larger ones (nx <= 12, nu <= 4) are handed to the library stated with 12
states and 4 controls (added states constant zero, added controls costing
u^2/2 and moving nothing: the solve is that of the plant as written) and take the
16-lanes-per-trajectory kernels of the quadrotors; arrays are widened on the
way in and cut back on the way out, so the Python interface keeps the plant's
own dimensions.  A plant that overrides
CalcDifferentialState (same dimension as the state; e.g. an angle difference
taken the short way round) has that method traced too.

A plant whose error state has another dimension than its state -- the
reference's own example is the quaternion quadrotor, control/models/Quadrotor_Quat.py:218-275
-- declares `ndx` and overrides, like that plant, `CalcDifferentialState`, `stage`, `final` and `Q_terms`.
All four are traced: they are written with `ilqr_amd.sym` and NumPy's `@` on the arrays handed in, and get their
raw derivatives from `self.derivatives(x, u)` / `self.final_derivatives(x)` (symbolic when x is, numbers otherwise):

    def stage(self, x, u):
        l_x, l_u, l_xx, l_ux, l_uu, A, B = self.derivatives(x, u)
        G = self.G(x)                                        # the plant's own nx x ndx map
        return l_x, l_u, l_xx, l_ux, l_uu, G.T @ A @ G, G.T @ B
    def Q_terms(self, x, l_x, l_u, l_xx, l_ux, l_uu, f_x, f_u, V_x, V_xx): ...   # as Quadrotor_Quat.py:267-275

(one trajectory per lane: nx + nu <= 8).
"""
from __future__ import annotations

import os
import types

import numpy as np

from .. import _lib
from .model import Model


class UserModel(Model):
    _model_id = _lib.MODEL_USER

    def __init__(self, nx: int, nu: int, params=None, name: str | None = None, ndx: int | None = None):
        super().__init__()
        self.nx, self.nu, self.ndx = int(nx), int(nu), int(nx if ndx is None else ndx)
        self.p = np.array([] if params is None else params, dtype=np.float64).ravel()
        self.name = name or type(self).__name__.lower()
        self._library = None
        cls = type(self)
        for meth in ("discrete_dynamics", "_cost_stage", "_cost_final"):
            if getattr(cls, meth) is getattr(Model, meth):
                raise TypeError(f"{cls.__name__} must define {meth} (README.md:25-31 of the reference)")
        self._user = (cls.discrete_dynamics, cls._cost_stage, cls._cost_final)
        # the plant's own CalcDifferentialState (model.py:22-23 overridden), traced like the other three
        self._user_diff = (cls.CalcDifferentialState
                           if cls.CalcDifferentialState is not UserModel.CalcDifferentialState else None)
        # the plant's own stage / final / Q_terms (Quadrotor_Quat.py:248-275): all three or none
        own = {k: getattr(cls, k) for k in ("stage", "final", "Q_terms") if getattr(cls, k) is not getattr(UserModel, k)}
        if own and len(own) != 3:
            raise TypeError(f"{cls.__name__} overrides {sorted(own)}: a plant that states its own Q-function terms "
                            "overrides stage, final and Q_terms together (Quadrotor_Quat.py:248-275)")
        if self.ndx != self.nx and (not own or self._user_diff is None):
            raise TypeError(f"{cls.__name__}: ndx = {self.ndx} != nx = {self.nx} needs the plant's own "
                            "CalcDifferentialState, stage, final and Q_terms (Quadrotor_Quat.py:218-275)")
        self._user_own = own or None
        self._sym_cache = None

    # ------------------------------------------------------------ the user's own Python
    def python_dynamics(self, x, u):
        return np.asarray(self._user[0](self, np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)),
                          dtype=np.float64)

    def python_cost_stage(self, x, u):
        return float(self._user[1](self, np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)))

    def python_cost_final(self, x):
        return float(self._user[2](self, np.asarray(x, dtype=np.float64)))

    # ------------------------------------------------------------ raw derivatives for the plant's own stage / final
    def _sym(self):
        """SymPy expressions of the three functions and their derivatives (in x, u, p symbols), built once."""
        if self._sym_cache is None:
            import sympy as sp
            from .. import codegen
            xs, us, ps, f, l, lf = codegen.user_expressions(self.nx, self.nu, self.p.shape[-1], self._traced(0),
                                                            self._traced(1), self._traced(2))[:6]
            lx = sp.Matrix([l]).jacobian(xs); lu = sp.Matrix([l]).jacobian(us); lfx = sp.Matrix([lf]).jacobian(xs)
            ex = dict(l_x=lx.T, l_u=lu.T, l_xx=lx.jacobian(xs), l_ux=lu.jacobian(xs), l_uu=lu.jacobian(us),
                      f_x=f.jacobian(xs), f_u=f.jacobian(us), lf_x=lfx.T, lf_xx=lfx.jacobian(xs))
            self._sym_cache = dict(xs=xs, us=us, ps=ps, ex=ex, fn={})
        return self._sym_cache

    def _derivs(self, keys, x, u):
        import sympy as sp
        c = self._sym()
        x = np.asarray(x)
        symbolic = x.dtype == object or (u is not None and np.asarray(u).dtype == object)
        p = np.asarray(self.p)
        out = []
        for k in keys:
            M = c["ex"][k]
            if symbolic:
                sub = dict(zip(c["xs"], x))
                if u is not None:
                    sub.update(zip(c["us"], np.asarray(u)))
                sub.update(zip(c["ps"], p.ravel()[:len(c["ps"])]) if p.size else [])
                a = np.array(M.xreplace(sub).tolist(), dtype=object)
            else:
                if k not in c["fn"]:
                    c["fn"][k] = sp.lambdify(list(c["xs"]) + list(c["us"]) + list(c["ps"]), M, "numpy")
                uu = np.zeros(self.nu) if u is None else np.asarray(u, dtype=np.float64)
                pp = p.ravel() if p.size else np.zeros(1)
                a = np.asarray(c["fn"][k](*np.asarray(x, dtype=np.float64), *uu, *pp[:len(c["ps"])]), dtype=np.float64)
            out.append(a.ravel() if a.shape[1] == 1 else a)
        return out

    def derivatives(self, x, u):
        """(l_x, l_u, l_xx, l_ux, l_uu, A, B) of the three functions the plant wrote, on the STATE (nx), for the
        plant's own stage(): SymPy entries when x or u hold symbols (tracing), numbers otherwise."""
        return tuple(self._derivs(("l_x", "l_u", "l_xx", "l_ux", "l_uu", "f_x", "f_u"), x, u))

    def final_derivatives(self, x):
        """(l_final_x, l_final_xx) on the state, for the plant's own final()."""
        return tuple(self._derivs(("lf_x", "lf_xx"), x, None))

    # ------------------------------------------------------------ tracing and building
    def _traced(self, which):
        fn = self._user[which]

        def call(*args):
            *xs, p = args
            saved = self.p
            self.p = p
            try:
                return fn(self, *xs)
            finally:
                self.p = saved
        return call

    def trace(self):
        """(xs, us, ps, f, l, lf): the three methods run once on SymPy symbols."""
        from .. import codegen
        diff = None
        if self._user_diff is not None:
            def diff(s, r, p):
                saved = self.p
                self.p = p
                try:
                    return self._user_diff(self, s, r)
                finally:
                    self.p = saved
        overrides = None
        if self._user_own:
            def with_p(fn):
                def call(*args):
                    *a, p = args
                    saved = self.p
                    self.p = p
                    try:
                        return fn(self, *a)
                    finally:
                        self.p = saved
                return call
            overrides = dict(stage=with_p(self._user_own["stage"]), final=with_p(self._user_own["final"]),
                             q_terms=with_p(self._user_own["Q_terms"]))
        return codegen.user_expressions(self.nx, self.nu, self.p.shape[-1], self._traced(0), self._traced(1),
                                        self._traced(2), diff, ndx=self.ndx, overrides=overrides)

    def header_text(self) -> str:
        from .. import codegen
        return codegen.generate_user_header(self.nx, self.nu, self.p.shape[-1], self.trace(), self.name)

    def library_dir(self, out_root=None) -> str:
        """Directory that holds (or will hold) libilqr_user.so and user_gen.h of this plant."""
        from .. import codegen
        return codegen.user_library_dir(codegen.user_key(self.nx, self.nu, self.p.shape[-1], self.trace()), self.name,
                                        out_root)

    def build(self, out_root=None, force=False):
        """Generate and compile (or find already compiled) the solver library for this plant and bind to it."""
        from .. import codegen
        self._library = codegen.build_user_library(self.nx, self.nu, self.p.shape[-1], self.trace(), self.name, out_root,
                                                   force=force)
        self._bind()
        return self

    def use_library(self, path: str):
        """Bind to a library built earlier for this same plant (build() on another machine, shipped with an app)."""
        if not os.path.exists(path):
            raise _lib.LibraryMissing(f"{path} not found: call build() where hipcc is available")
        self._library = path
        self._bind()
        return self

    def _bind(self):
        # from here on the Model interface evaluates the compiled code (stage / final of a plant that states its own
        # Q-function terms come back as the library has them: cost derivatives already on the error state, i.e. the
        # plant's Q_terms at V = 0, the convention of the quaternion quadrotor in include/ilqr_hip.h)
        for meth in ("discrete_dynamics", "_cost_stage", "_cost_final") + (("stage", "final") if self._user_own else ()):
            setattr(self, meth, types.MethodType(getattr(Model, meth), self))

    # ------------------------------------------------------------ plants the library states wider than written
    def _wide(self):
        from .. import codegen
        return codegen.is_wide(self.nx, self.nu)

    def _widen(self, a, have, want):
        if not self._wide():
            return a
        import torch
        if torch.is_tensor(a):
            pad = torch.zeros(a.shape[:-1] + (want - have,), dtype=a.dtype, device=a.device)
            return torch.cat([a, pad], dim=-1)
        a = np.asarray(a, dtype=np.float64)
        return np.concatenate([a, np.zeros(a.shape[:-1] + (want - have,))], axis=-1)

    def _lib_x(self, x):
        from .. import codegen
        return self._widen(x, self.nx, codegen.WIDE_NX)

    def _lib_u(self, u):
        from .. import codegen
        return self._widen(u, self.nu, codegen.WIDE_NU)

    def _own(self, key, a):
        if not self._wide():
            return a
        n, m = self.nx, self.nu
        cut = dict(x=(n,), xn=(n,), u=(m,), l_x=(n,), l_u=(m,), l_xx=(n, n), l_ux=(m, n), l_uu=(m, m), f_x=(n, n),
                   f_u=(n, m), V_x=(n,), V_xx=(n, n)).get(key)
        if cut is None:
            return a
        if len(cut) == 1:
            return a[..., :cut[0]]
        return a[..., :cut[0], :cut[1]]

    def _param_block(self):
        if self._library is None:
            raise RuntimeError(f"{type(self).__name__}: call build() (or use_library()) before solving")
        return self.p if self.p.size else np.zeros(1)

    def set_params(self, params):
        """New values for the numbers declared as `params`: [np] for the whole batch or [B, np], one set per
        trajectory of the batches solved from now on (batched goals; no rebuild either way)."""
        p = np.asarray(params, dtype=np.float64)
        n = self.p.shape[-1]
        if p.ndim == 2 and p.shape[1] == n and n > 0:
            self.p = np.ascontiguousarray(p)
            return
        p = p.ravel()
        if p.size != n:
            raise ValueError(f"{type(self).__name__} was traced with {n} parameters, got {p.size}")
        self.p = p
