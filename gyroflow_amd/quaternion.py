"""Quaternion helpers under the reference's names and (w, x, y, z) convention (/root/reference/quaternion.py:4-137).

Host-side conveniences only: the per-sample recurrences of the quaternion pipeline run in csrc/quat_host.cpp; these numpy
versions serve callers that want single operations and the tests that cross-check the C code."""
import numpy as np

_DOT_THRESHOLD = 0.9995        # above this slerp degenerates to a normalised lerp (quaternion.py:97,118)


def quaternion(w, x, y, z):
    return np.array([w, x, y, z])


def vector(x, y, z):
    return np.array([x, y, z])


def normalize(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.sqrt(np.dot(q, q))


def quaternion_multiply(Q1, Q2):
    """Hamilton product Q1 * Q2."""
    a = np.asarray(Q1, dtype=np.float64)
    b = np.asarray(Q2, dtype=np.float64)
    s = a[0] * b[0] - np.dot(a[1:], b[1:])
    v = a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:])
    return np.concatenate(([s], v))


def inverse(q):
    """Inverse of a unit quaternion: the conjugate."""
    q = np.asarray(q, dtype=np.float64)
    return q * np.array([1.0, -1.0, -1.0, -1.0])


conjugate = inverse


def rotate_vector(q, v):
    p = np.concatenate(([0.0], np.asarray(v, dtype=np.float64)))
    return quaternion_multiply(quaternion_multiply(q, p), conjugate(q))[1:]


rotate_vector_fast = rotate_vector


def rot_between(q1, q2):
    """Rotation that takes orientation q1 to orientation q2."""
    return quaternion_multiply(inverse(q1), q2)


def slerp(v0, v1, t_array):
    """Spherical interpolation from v0 to v1 at every t of t_array -> [len(t), 4]."""
    a = np.asarray(v0, dtype=np.float64)
    b = np.asarray(v1, dtype=np.float64)
    t = np.atleast_1d(np.asarray(t_array, dtype=np.float64))[:, None]
    c = float(np.dot(a, b))
    if c < 0.0:
        b, c = -b, -c
    if c > _DOT_THRESHOLD:
        r = a[None, :] + t * (b - a)[None, :]
        return r / np.linalg.norm(r, axis=1, keepdims=True)
    ang = np.arccos(c)
    part = np.sin(ang * t) / np.sin(ang)
    return (np.cos(ang * t) - c * part) * a[None, :] + part * b[None, :]


def single_slerp(v0, v1, t):
    return slerp(v0, v1, [t])[0]


def angle_between(q1, q2):
    return 2 * np.arccos(min(rot_between(q1, q2)[0], 1))
