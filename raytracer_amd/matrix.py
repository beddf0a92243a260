"""Transform builders of the host API: the `CGMatrix` trait (src/matrix.rs:4-89) and
`view_transformation` (src/tuple.rs:50-60), in f32 like the reference.

Matrices are 4x4 numpy float32 arrays in the mathematical (row, column) sense; `to_abi` lays them out
column-major, the way nalgebra stores `Matrix4<f32>` and include/rt355.h expects them.
"""
import numpy as np

F = np.float32


def identity():
    return np.eye(4, dtype=F)


def rotation_x(x):
    c, s = F(np.cos(F(x))), F(np.sin(F(x)))
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=F)


def rotation_y(x):
    c, s = F(np.cos(F(x))), F(np.sin(F(x)))
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=F)


def rotation_z(x):
    c, s = F(np.cos(F(x))), F(np.sin(F(x)))
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=F)


def scaling(x, y, z):
    return np.diag(np.array([x, y, z, 1], dtype=F)).astype(F)


def translation(x, y, z):
    m = np.eye(4, dtype=F)
    m[0, 3], m[1, 3], m[2, 3] = F(x), F(y), F(z)
    return m


translate = translation  # src/matrix.rs:85-89 keeps both names


def point(x, y, z):
    return np.array([x, y, z, 1], dtype=F)


def vector(x, y, z):
    return np.array([x, y, z, 0], dtype=F)


def _normalize(v):
    return (v / F(np.sqrt(np.dot(v, v)))).astype(F)


def cross4(a, b):
    return vector(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def reflect(v, n):
    """`TupleOperation::reflect` (src/tuple.rs:38-40)."""
    v = np.asarray(v, dtype=F)
    n = np.asarray(n, dtype=F)
    return (v - n * F(2) * F(np.dot(v, n))).astype(F)


def view_transformation(frm, to, up):
    """`from.view_transformation(&to, &up)` (src/tuple.rs:50-60)."""
    frm = np.asarray(frm, dtype=F)
    to = np.asarray(to, dtype=F)
    up = np.asarray(up, dtype=F)
    forward = _normalize(to - frm)
    up_normal = _normalize(up)
    left = cross4(forward, up_normal)
    true_up = cross4(left, forward)
    orientation = np.array(
        [
            [left[0], left[1], left[2], 0],
            [true_up[0], true_up[1], true_up[2], 0],
            [-forward[0], -forward[1], -forward[2], 0],
            [0, 0, 0, 1],
        ],
        dtype=F,
    )
    return (orientation @ translation(-frm[0], -frm[1], -frm[2])).astype(F)


def to_abi(m):
    """Column-major flat list of 16 floats."""
    return np.asarray(m, dtype=F).T.reshape(16)
