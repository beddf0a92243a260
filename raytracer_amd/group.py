"""Scene tree and its flattening: `Scene::to_transformed_shapes` (src/group.rs:117-168).

This runs on the host BEFORE the render path: a `World` only ever holds a flat list in which Group
entries are inert (src/shape.rs:376).  The flattening keeps the reference's semantics:

  * only the IMMEDIATE parent group's transformation is composed into a child
    (`group_transformation * shape.transformation()`, src/group.rs:146); transforms of outer groups do
    not accumulate, because the recursion re-reads the child's original transformation;
  * the root shape and every child (groups included) is pushed to the output in depth-first order;
  * a non-group node may have children; they pass through unchanged (src/group.rs:154-166).

`accumulate=True` is the evident intent instead (full parent chain composed), offered for new scenes.
"""
import numpy as np

from . import _abi


class Node:
    def __init__(self, shape, children=None):
        self.shape = shape
        self.parent = None
        self.children = []
        for ch in children or []:
            self.add(ch)

    def add(self, child):
        node = child if isinstance(child, Node) else Node(child)
        node.parent = self
        self.children.append(node)
        return node


def _inverse(node):
    return np.linalg.inv(node.shape.transformation.astype(np.float32)).astype(np.float32)


def world_to_object(node, point):
    """`Scene::world_to_object` (src/group.rs:94-103): the point is taken through the inverse transformation of
    every ancestor, outermost first, then through the node's own.  f32 like the reference."""
    point = np.asarray(point, dtype=np.float32)
    if node.parent is not None:
        point = world_to_object(node.parent, point)
    return _inverse(node) @ point


def normal_to_world(node, normal):
    """`Scene::normal_to_world` (src/group.rs:105-115): transposed inverse, w cleared, normalised, at every level
    from the node up to the root."""
    n = _inverse(node).T @ np.asarray(normal, dtype=np.float32)
    n[3] = np.float32(0)
    n = (n / np.float32(np.sqrt(np.float32(n @ n)))).astype(np.float32)
    return n if node.parent is None else normal_to_world(node.parent, n)


def normal(node, world_point, local_normal=None):
    """`Scene::normal` (src/group.rs:87-92).  `local_normal(object_point)` defaults to the sphere's
    (src/sphere.rs: point - origin), the shape the reference's own test uses."""
    p = world_to_object(node, world_point)
    if local_normal is None:
        if node.shape.kind != _abi.SPHERE:
            raise NotImplementedError("pass local_normal for shapes other than spheres")
        n = p - np.array([0, 0, 0, 1], dtype=np.float32)
    else:
        n = np.asarray(local_normal(p), dtype=np.float32)
    return normal_to_world(node, n)


def to_transformed_shapes(root, accumulate=False):
    out = [root.shape.copy()]

    def fold(node, carried):
        if node.shape.kind == _abi.GROUP:
            gt = node.shape.transformation if not accumulate else carried
            for ch in node.children:
                s = ch.shape.copy()
                s.set_transform((gt @ ch.shape.transformation).astype(np.float32))
                out.append(s)
                fold(ch, s.transformation if accumulate else None)
        else:
            for ch in node.children:
                out.append(ch.shape.copy())
                fold(ch, ch.shape.transformation if accumulate else None)

    fold(root, root.shape.transformation)
    return out
