"""Unit-quaternion helpers on NumPy arrays, scalar first: q = [w, x, y, z].

Same names, arguments and results as maths/quaternions.py:8-81 of the reference (pinned by
tests/golden/quat_utils.npz, which holds the reference's own outputs), written from the algebra:
the product p (x) q is L(p) q with L(p) = [[w, -v'], [v, w I + hat(v)]], the attitude Jacobian is the
last three columns of L(q), and the rotation matrix is the vector block of L(q) R(q)' .
"""
from __future__ import annotations

import numpy as np

Quaternion = np.ndarray


def hat(q) -> np.ndarray:
    """Skew-symmetric matrix of the vector part of q: hat(q) @ a = v x a  (quaternions.py:8-18)."""
    x, y, z = (float(c) for c in np.asarray(q)[1:].squeeze())
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def GetConjugate(q) -> np.ndarray:
    return np.asarray(q) * np.array([1, -1, -1, -1])           # quaternions.py:20-21


def GetLeftMatrix(q) -> np.ndarray:
    """L(q) with L(q) p = q (x) p  (quaternions.py:24-38)."""
    q = np.asarray(q)
    w, v = q[0], q[1:].squeeze()
    out = np.empty((4, 4), dtype=q.dtype)
    out[0, 0] = w
    out[0, 1:] = -v
    out[1:, 0] = v
    out[1:, 1:] = w * np.eye(3) + hat(q)
    return out


def GetAttitudeJacobian(q):
    """G(q) [4 x 3]: d q / d (three-parameter rotation error) (quaternions.py:40-47)."""
    return GetLeftMatrix(q)[:, 1:]


def ParamToQuaternion(phi):
    """Rodrigues parameters -> quaternion, shape (1, 4) like the reference (quaternions.py:49-53)."""
    phi = np.asarray(phi, dtype=np.float64).reshape(1, 3)
    return np.hstack((np.ones((1, 1)), phi)) / np.sqrt(1.0 + float((phi * phi).sum()))


def QuaternionToParam(q: Quaternion) -> np.ndarray:
    """Quaternion -> Rodrigues parameters; q is normalised IN PLACE first, as in quaternions.py:55-63."""
    q /= np.linalg.norm(q)
    return q[1:] / q[0]


def QuaternionToRotation(q: Quaternion):
    """Rotation matrix of q (quaternions.py:65-70)."""
    w, v = float(np.asarray(q)[0]), np.asarray(q, dtype=np.float64)[1:].squeeze()
    K = hat(q)
    return np.outer(v, v) + w * w * np.eye(3) + 2.0 * w * K + K @ K


def SampleQuaternion(near_identity=False):
    """A random unit quaternion [w, x, y, z]; near the identity: Euler angles ~ 0.5 N(0, 1)
    (quaternions.py:73-81; the reference assembles [w, y, z, w] from SciPy's scalar-last vector there, which
    is not a unit quaternion -- this returns the rotation it sampled)."""
    from scipy.spatial.transform import Rotation
    rot = Rotation.from_euler("xyz", 0.5 * np.random.randn(3)) if near_identity else Rotation.random()
    x, y, z, w = rot.as_quat()
    return np.array([w, x, y, z])
