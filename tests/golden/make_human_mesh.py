"""Generates tests/golden/human_mesh.npz - the surface SURVEY.md 8(d) samples its benchmark clouds from.  Run HERE (the
container that has /root/reference); the GPU box only ever reads the .npz.

  python tests/golden/make_human_mesh.py

Source: /root/reference/load/shapes/human.obj (1629 vertices, 1694 polygons: the stand-in for the SMPL-X body
HumanGaussian initialises from - the SMPL-X model files are not in the tree).  Normalised exactly as the reference
normalises its body mesh before sampling it (threestudio/utils/poser.py:337-346: centre of the bounding box to the origin,
scale 0.6 / largest extent; :349-352: swap y and z (OpenGL -> Blender); threestudio/systems/GaussianDreamer.py:122
`skel.scale(-10)` -> poser.py:354-357: x 1.1^10), polygons fan-triangulated.  The sampling itself (area-uniform, seeded) is
humangaussian_amd/synth.py::human_points, so any point count comes from this one 60 KB file.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/load/shapes/human.obj"


def main():
    verts, tris = [], []
    for line in open(SRC):
        if line.startswith("v "):
            verts.append([float(x) for x in line.split()[1:4]])
        elif line.startswith("f "):
            idx = [int(tok.split("/")[0]) - 1 for tok in line.split()[1:]]
            for k in range(1, len(idx) - 1):                      # fan triangulation of the quads / pentagons
                tris.append([idx[0], idx[k], idx[k + 1]])
    v = np.asarray(verts, np.float64)
    f = np.asarray(tris, np.int32)
    vmin, vmax = v.min(0), v.max(0)
    v = (v - (vmax + vmin) / 2) * (0.6 / np.max(vmax - vmin))     # poser.py:337-346
    v[:, [1, 2]] = v[:, [2, 1]]                                   # poser.py:349-352
    v *= 1.1 ** 10                                                # GaussianDreamer.py:122 -> poser.py:354-357
    out = os.path.join(HERE, "human_mesh.npz")
    np.savez_compressed(out, vertices=v.astype(np.float32), faces=f,
                        source="load/shapes/human.obj, normalised as threestudio/utils/poser.py:337-357 with scale(-10)")
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum()
    print(out, "vertices", v.shape, "triangles", f.shape, "extent", v.max(0) - v.min(0), "area %.3f" % area)


if __name__ == "__main__":
    main()
