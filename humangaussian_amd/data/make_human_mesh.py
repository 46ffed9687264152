"""Generates humangaussian_amd/data/human_mesh.npz - the surface SURVEY.md 8(d) samples its benchmark clouds from.  The
.npz is a LOCAL, git-ignored artefact (the mesh is a third-party asset of the reference tree: it is not redistributed with
this repository); `__graft_entry__.build()` runs this script wherever /root/reference exists, the file then travels to the
GPU box with the working tree like the built .so files.  Without it every consumer falls back to the analytic capsule
humanoid (`synth.humanoid_points` / `synth.humanoid_mesh`) and says so.

  python humangaussian_amd/data/make_human_mesh.py

Source: /root/reference/load/shapes/human.obj (1629 vertices, 1694 polygons: the stand-in for the SMPL-X body
HumanGaussian initialises from - the SMPL-X model files are not in the tree).  Normalised exactly as the reference
normalises its body mesh before sampling it (threestudio/utils/poser.py:337-346: centre of the bounding box to the origin,
scale 0.6 / largest extent; :349-352: swap y and z (OpenGL -> Blender); threestudio/systems/GaussianDreamer.py:122
`skel.scale(-10)` -> poser.py:354-357: x 1.1^10), polygons fan-triangulated.  The sampling itself (area-uniform, seeded) is
humangaussian_amd/synth.py::human_points, so any point count comes from this one 30 KB file.
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
    out = os.path.join(HERE, "human_mesh.npz")      # HERE = humangaussian_amd/data
    np.savez_compressed(out, vertices=v.astype(np.float32), faces=f,
                        source="load/shapes/human.obj, normalised as threestudio/utils/poser.py:337-357 with scale(-10)")
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum()
    print(out, "vertices", v.shape, "triangles", f.shape, "extent", v.max(0) - v.min(0), "area %.3f" % area)


if __name__ == "__main__":
    main()
