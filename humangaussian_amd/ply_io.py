"""PLY checkpoint I/O, byte-compatible with the reference's `GaussianModel.save_ply` /
`load_ply` (/root/reference/gaussiansplatting/scene/gaussian_model.py:187-266): one
`vertex` element, binary little-endian, all properties `float` in the order
x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_*; SH coefficients are stored
channel-major (the reference transposes (P, M, 3) -> (P, 3, M) before flattening); opacity
is the pre-sigmoid logit, scales are log-scales, rotations un-normalised (w, x, y, z).
Pure numpy (the reference uses the `plyfile` package, absent here; the header this module
writes is exactly what plyfile emits for the same structured array)."""
from __future__ import annotations

import numpy as np


def attribute_names(num_rest: int):
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(num_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    return names


def save_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """Arrays are the model's RAW parameters: xyz (P,3), features_dc (P,1,3),
    features_rest (P,M-1,3), opacity (P,1) logits, scaling (P,3) log-scales, rotation (P,4)."""
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))  # noqa: E731
    xyz, opacity, scaling, rotation = f32(xyz), f32(opacity).reshape(-1, 1), f32(scaling), f32(rotation)
    P = xyz.shape[0]
    f_dc = f32(features_dc).reshape(P, -1, 3).transpose(0, 2, 1).reshape(P, -1)
    f_rest = f32(features_rest).reshape(P, -1, 3).transpose(0, 2, 1).reshape(P, -1)
    names = attribute_names(f_rest.shape[1])
    rows = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, opacity, scaling, rotation], axis=1)
    assert rows.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {P}\n" + \
        "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rows.astype("<f4").tobytes())


def load_ply(path, max_sh_degree=None, kiui_axes=False):
    """Returns a dict of RAW parameter arrays shaped like the reference's nn.Parameters:
    xyz (P,3), features_dc (P,1,3), features_rest (P,M-1,3), opacity (P,1), scaling (P,3),
    rotation (P,4), plus max_sh_degree.

    kiui_axes=True applies the fix-ups of the animation-side loader
    (/root/reference/gs_renderer.py:576-581): y and z of the positions and of the scales are
    swapped, the quaternion's y and z components are swapped and its w is negated."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii").splitlines()
    if lines[0] != "ply" or "binary_little_endian" not in lines[1]:
        raise ValueError("expected a binary little-endian PLY")
    P, names = None, []
    for ln in lines[2:]:
        tok = ln.split()
        if tok[:2] == ["element", "vertex"]:
            P = int(tok[2])
        elif tok and tok[0] == "property":
            if tok[1] not in ("float", "float32"):
                raise ValueError(f"unsupported property type {tok[1]}")
            names.append(tok[2])
    rows = np.frombuffer(data, dtype="<f4", count=P * len(names), offset=end).reshape(P, len(names))
    col = {n: i for i, n in enumerate(names)}
    take = lambda ns: np.stack([rows[:, col[n]] for n in ns], axis=1)  # noqa: E731
    rest = sorted((n for n in names if n.startswith("f_rest_")), key=lambda n: int(n.split("_")[-1]))
    deg = int(round(math_sqrt((len(rest) + 3) / 3))) - 1
    if max_sh_degree is not None and len(rest) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise AssertionError("f_rest count does not match max_sh_degree")
    scale_names = sorted((n for n in names if n.startswith("scale_")), key=lambda n: int(n.split("_")[-1]))
    rot_names = sorted((n for n in names if n.startswith("rot")), key=lambda n: int(n.split("_")[-1]))
    f_dc = take(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(P, 3, 1).transpose(0, 2, 1)
    f_rest = (take(rest).reshape(P, 3, -1).transpose(0, 2, 1) if rest else np.zeros((P, 0, 3), np.float32))
    xyz, scaling, rotation = take(["x", "y", "z"]), take(scale_names), take(rot_names)
    if kiui_axes:
        xyz[:, [1, 2]] = xyz[:, [2, 1]]               # coordinate shift
        scaling[:, [1, 2]] = scaling[:, [2, 1]]
        rotation[:, [2, 3]] = rotation[:, [3, 2]]     # (w, x, y, z) -> (w, x, z, y) ...
        rotation[:, [0]] *= -1                        # ... and the handedness flip
    return dict(xyz=xyz, features_dc=np.ascontiguousarray(f_dc),
                features_rest=np.ascontiguousarray(f_rest), opacity=take(["opacity"]),
                scaling=scaling, rotation=rotation, max_sh_degree=deg)


def math_sqrt(x):
    return float(np.sqrt(x))
