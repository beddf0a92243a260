"""OBJ loader feeding the scene uploader: the step right before the render path
(`Wavefront::parse`, src/wavefront.rs:39-184).

Like the reference it understands `v`, `vn`, `g` and `f` lines, emits SmoothTriangle shapes only, uses
1-based indices (slot 0 of `vertices` / `normals` is a dummy, src/wavefront.rs:48-49) and treats a
`f` line whose first vertex contains `/` as `v/vt/vn`.

Polygons with more than 3 vertices are fanned over the FACE's own vertex list (the evident intent).
The reference instead fans over the file-global vertex indices 1, index, index+1
(src/wavefront.rs:101-116, :143-155), which only coincides with the intent for faces that are
literally `1 2 3 4 ...`; `ref_parser_quirk=True` reproduces that behaviour.
"""
import numpy as np

from .host import Shape


class Wavefront:
    def __init__(self):
        self.vertices = [(0.0, 0.0, 0.0)]
        self.normals = [(0.0, 0.0, 0.0)]
        self.groups = []
        self.comments = []
        self.shapes = []

    @classmethod
    def parse(cls, path, ref_parser_quirk=False, material=None, transformation=None):
        w = cls()
        with open(path, "r") as fh:
            for line in fh:
                words = line.split()
                if not words or len(words[0]) > 2:
                    w.comments.append(line)
                    continue
                cat = words[0]
                if cat == "v":
                    w.vertices.append(tuple(float(np.float32(x)) for x in words[1:4]))
                elif cat == "vn":
                    w.normals.append(tuple(float(np.float32(x)) for x in words[1:4]))
                elif cat == "g":
                    w.groups.append(line)
                elif cat == "f":
                    w._face(words[1:], ref_parser_quirk, material, transformation)
                else:
                    w.comments.append(line)
        return w

    def _face(self, words, quirk, material, transformation):
        with_normals = "/" in words[0]
        vs, ns = [], []
        for word in words:
            if with_normals:
                f = word.split("/")
                vs.append(self.vertices[int(f[0])])
                # `v/vt` (no normal index) is read as a vertex without a normal; the reference indexes f[2]
                # unconditionally (src/wavefront.rs:93) and panics on such a file
                ns.append(self.normals[int(f[2])] if len(f) >= 3 and f[2] else (0.0, 0.0, 0.0))
            else:
                vs.append(self.vertices[int(word)])
                ns.append((0.0, 0.0, 0.0))
        l = len(vs)
        if l < 3:
            return
        if l == 3:
            tris = [((vs[0], vs[1], vs[2]), (ns[0], ns[1], ns[2]))]
        elif quirk:
            zero = (0.0, 0.0, 0.0)
            tris = []
            for index in range(2, l):
                p = (self.vertices[1], self.vertices[index], self.vertices[index + 1])
                n = (self.normals[1], self.normals[index], self.normals[index + 1]) if with_normals else (zero, zero, zero)
                tris.append((p, n))
        else:
            tris = [((vs[0], vs[i], vs[i + 1]), (ns[0], ns[i], ns[i + 1])) for i in range(1, l - 1)]
        for p, n in tris:
            mat = material.copy() if material is not None else None
            self.shapes.append(Shape.SmoothTriangle(transformation, mat, p[0], p[1], p[2], n[0], n[1], n[2]))
