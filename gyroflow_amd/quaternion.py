"""Quaternion helpers with the reference's names and conventions (/root/reference/quaternion.py): (w,x,y,z) numpy arrays."""
import numpy as np


def quaternion(w, x, y, z):
    return np.array([w, x, y, z])


def vector(x, y, z):
    return np.array([x, y, z])


def normalize(q):
    return q / np.sqrt(q.dot(q))


def quaternion_multiply(Q1, Q2):
    w0, x0, y0, z0 = Q2
    w1, x1, y1, z1 = Q1
    return np.array([-x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0,
                     x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
                     -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
                     x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0])


def inverse(q):
    return quaternion(q[0], -q[1], -q[2], -q[3])


conjugate = inverse


def rot_between(q1, q2):
    """Rotation quaternion from q1 to q2."""
    return quaternion_multiply(inverse(q1), q2)


def single_slerp(v0, v1, t):
    v0 = np.array(v0)
    v1 = np.array(v1)
    dot = np.sum(v0 * v1)
    if dot < 0.0:
        v1 = -v1
        dot = -dot
    if dot > 0.9995:
        result = v0 + t * (v1 - v0)
        return result / np.linalg.norm(result)
    theta_0 = np.arccos(dot)
    sin_theta_0 = np.sin(theta_0)
    theta = theta_0 * t
    sin_theta = np.sin(theta)
    s0 = np.cos(theta) - dot * sin_theta / sin_theta_0
    s1 = sin_theta / sin_theta_0
    return (s0 * v0) + (s1 * v1)


def angle_between(q1, q2):
    z = quaternion_multiply(inverse(q1), q2)
    return 2 * np.arccos(min(z[0], 1))
