"""Math functions for writing a plant once and using it both with numbers and with symbols.

The reference's car model picks its math module from the type of the first
state component (control/models/car.py:67-69: `m = sym if x.dtype == object else
np`) so that one `discrete_dynamics` serves NumPy evaluation and symbolic
differentiation.  The functions here do the same per call: SymPy when any
argument is a SymPy expression, NumPy otherwise.

    from ilqr_amd import sym as m
    heading = m.atan2(m.sin(x[2]), m.cos(x[2]))
"""
from __future__ import annotations

import numpy as np
import sympy as sp


def _symbolic(*args):
    for a in args:
        if isinstance(a, sp.Basic):
            return True
        if isinstance(a, np.ndarray) and a.dtype == object:
            return True
    return False


def _unary(np_fn, sp_fn):
    def f(a):
        if _symbolic(a):
            if isinstance(a, np.ndarray):
                return np.array([sp_fn(e) for e in a.ravel()], dtype=object).reshape(a.shape)
            return sp_fn(a)
        return np_fn(a)
    f.__name__ = np_fn.__name__
    return f


sin = _unary(np.sin, sp.sin)
cos = _unary(np.cos, sp.cos)
tan = _unary(np.tan, sp.tan)
exp = _unary(np.exp, sp.exp)
log = _unary(np.log, sp.log)
sqrt = _unary(np.sqrt, sp.sqrt)
tanh = _unary(np.tanh, sp.tanh)
sinh = _unary(np.sinh, sp.sinh)
cosh = _unary(np.cosh, sp.cosh)
atan = _unary(np.arctan, sp.atan)
asin = _unary(np.arcsin, sp.asin)
acos = _unary(np.arccos, sp.acos)


def atan2(y, x):
    return sp.atan2(y, x) if _symbolic(y, x) else np.arctan2(y, x)


def array(seq):
    """np.array that keeps SymPy entries as objects and numbers as float64."""
    seq = list(seq)
    return np.array(seq, dtype=object) if _symbolic(*seq) else np.array(seq, dtype=np.float64)
