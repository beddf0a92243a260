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
        self.children = list(children) if children else []

    def add(self, child):
        node = child if isinstance(child, Node) else Node(child)
        self.children.append(node)
        return node


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
