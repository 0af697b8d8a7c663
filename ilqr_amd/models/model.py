"""Plant-model base class: the reference's `Model` interface on top of libilqr_hip.

Mirrors control/models/model.py:3-53 (same method names, argument meaning and
return shapes).  The arithmetic of every method runs on the GPU through the C
ABI (`ilqr_model_eval`, `ilqr_rollout_cost`); a model is identified to the
library by `_model_id` and its parameter block `_param_block()`.

The four shipped plants use the generated device code inside libilqr_hip.so.
A plant written by the user (the reference's README.md:25-31 workflow: subclass
Model, write discrete_dynamics / _cost_stage / _cost_final) is supported through
`ilqr_amd.models.user.UserModel`, which traces those three functions into device
code and carries the path of its own solver library in `_library`.  A subclass
that merely overrides Python methods of a built-in plant is not traced and is
rejected by `iLQR` with a TypeError.
"""
from __future__ import annotations

import numpy as np


class Model:
    _model_id = None  # set by the built-in plants

    def __init__(self):
        self.nx = None
        self.nu = None
        self.ndx = None

    # ------------------------------------------------------------ plumbing
    def _param_block(self) -> np.ndarray:
        raise NotImplementedError

    def _horizon(self):
        return getattr(self, "N", None)

    def _evaluator(self, N=2):
        """A BatchSolver bound to the current parameters (cached per (N, params))."""
        from ..batch import BatchSolver
        par = np.ascontiguousarray(self._param_block(), dtype=np.float64)
        if par.ndim != 1:
            raise ValueError("this plant holds one parameter set per trajectory: the single-point Model interface "
                             "needs one set (evaluate batches through ilqr_amd.batch.BatchSolver)")
        key = (int(N), par.tobytes())
        cache = self.__dict__.setdefault("_eval_cache", {})
        if key not in cache:
            cache.clear()
            cache[key] = BatchSolver(self._model_id, par, int(N), 1, 1, 0.0, library=getattr(self, "_library", None))
        return cache[key]

    # The solver library may state the plant with more states / controls than the plant as written (user plants with
    # nx + nu > 8, ilqr_amd/models/user.py): arrays go in through _lib_x / _lib_u and come back through _own.
    def _lib_x(self, x):
        return x

    def _lib_u(self, u):
        return u

    def _own(self, key, a):
        return a

    def _eval(self, x, u, what):
        x = np.asarray(x, dtype=np.float64).reshape(1, self.nx)
        u = np.zeros((1, self.nu)) if u is None else np.asarray(u, dtype=np.float64).reshape(1, self.nu)
        out = self._evaluator().model_eval(self._lib_x(x), self._lib_u(u), what)
        return [self._own(k, out[k][0].cpu().numpy()) for k in what]

    # ------------------------------------------------------------ interface
    def get_dims(self):
        return self.nx, self.nu, self.ndx                       # model.py:9-10

    def discrete_dynamics(self, x, u):
        return self._eval(x, u, ("xn",))[0]                     # model.py:12

    def rollout(self, x0, u_trj):
        """model.py:15-20: x_trj [len(u_trj)+1, nx]."""
        u_trj = np.asarray(u_trj, dtype=np.float64)
        x, _ = self._evaluator(u_trj.shape[0] + 1).rollout_cost(self._lib_x(np.asarray(x0, dtype=np.float64)[None]),
                                                                 self._lib_u(u_trj[None]))
        return self._own("x", x[0].cpu().numpy())

    def CalcDifferentialState(self, state, ref_state):
        return np.asarray(state) - np.asarray(ref_state)       # model.py:22-23

    def _cost_stage(self, x, u, *ignored_refs):
        return float(self._eval(x, u, ("cs",))[0])              # model.py:25

    def _cost_final(self, x, *ignored_refs):
        return float(self._eval(x, None, ("cf",))[0])           # model.py:28

    def cost_trj(self, x_trj, u_trj):
        """model.py:31-37: sum of stage costs along the given trajectory plus the final cost."""
        from ..batch import BatchSolver  # noqa: F401  (fails loudly if the library is missing)
        x_trj = np.asarray(x_trj, dtype=np.float64)
        u_trj = np.asarray(u_trj, dtype=np.float64)
        ev = self._evaluator()
        S = u_trj.shape[0]
        out = ev.model_eval(self._lib_x(x_trj[:S]), self._lib_u(u_trj), ("cs",))["cs"].cpu().numpy()
        total = 0.0
        for c in out:
            total += float(c)
        total += self._cost_final(x_trj[-1])
        return total

    def stage(self, x, u):
        """model.py:39: (l_x, l_u, l_xx, l_ux, l_uu, f_x, f_u)."""
        return tuple(self._eval(x, u, ("l_x", "l_u", "l_xx", "l_ux", "l_uu", "f_x", "f_u")))

    def _linearize_discrete(self, ref_state, ref_action):
        """(A, B) of the integrator step at (ref_state, ref_action): Quadrotor_EA.py:202-204, acrobot.py
        likewise; for the quaternion plant on the error state, G'AG and G'B (Quadrotor_Quat.py:231-238).
        A scalar ref_action stands for that value in every channel (the reference's controllers pass 0)."""
        u = np.array(np.broadcast_to(np.asarray(ref_action, dtype=np.float64), (self.nu,)))
        f_x, f_u = self._eval(ref_state, u, ("f_x", "f_u"))
        return f_x, f_u

    def _linearize_continuous(self, ref_state, ref_action):
        """(A, B) of the continuous-time dynamics at (ref_state, ref_action): acrobot.py:148-150,
        Quadrotor_EA.py:206-208; for the quaternion plant on the error state, G'AG and G'B
        (Quadrotor_Quat.py:240-246).  The car has none (car.py states the step)."""
        u = np.array(np.broadcast_to(np.asarray(ref_action, dtype=np.float64), (self.nu,)))
        x = np.asarray(ref_state, dtype=np.float64).reshape(1, self.nx)
        A, B = self._evaluator().linearize_continuous(x, u.reshape(1, self.nu))
        return A[0].cpu().numpy(), B[0].cpu().numpy()

    def final(self, x):
        """model.py:42: (V_x, V_xx)."""
        return tuple(self._eval(x, None, ("V_x", "V_xx")))

    def Q_terms(self, x, l_x, l_u, l_xx, l_ux, l_uu, f_x, f_u, V_x, V_xx):
        """model.py:45-53 (host arithmetic on the small matrices handed in)."""
        Q_x = l_x + f_x.T @ V_x
        Q_u = l_u + f_u.T @ V_x
        Q_xx = l_xx + f_x.T @ V_xx @ f_x
        Q_ux = l_ux + f_u.T @ V_xx @ f_x
        Q_uu = l_uu + f_u.T @ V_xx @ f_u
        return Q_x, Q_u, Q_xx, Q_ux, Q_uu
