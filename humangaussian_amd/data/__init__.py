"""Optional local assets of the benchmarks and the animation leg (nothing here is needed by the rasterizer itself).

  human_mesh.npz            the reference's load/shapes/human.obj, normalised (make_human_mesh.py)
  amass_test_17_poses.npz   the reference's content/amass_test_17.npz poses (make_motion.py)

Both are third-party data of the reference tree and are NOT redistributed with this repository: the .npz files are
git-ignored, built locally by the two scripts where /root/reference exists (`__graft_entry__.build()` does it), and every
consumer has a procedural fallback (`synth.humanoid_points` / `synth.humanoid_mesh`, `animation.MotionDriver`'s sway) that
it reports in its workload description.
"""
import os
import subprocess
import sys

DIR = os.path.dirname(os.path.abspath(__file__))
HUMAN_MESH = os.path.join(DIR, "human_mesh.npz")
MOTION = os.path.join(DIR, "amass_test_17_poses.npz")


def have_human_mesh() -> bool:
    return os.path.exists(HUMAN_MESH)


def have_motion() -> bool:
    return os.path.exists(MOTION)


def build_if_possible(reference_root: str = "/root/reference", verbose: bool = False) -> None:
    """Generate the missing assets from the reference tree (a no-op where it does not exist, e.g. on the GPU box)."""
    jobs = ((HUMAN_MESH, "make_human_mesh.py", os.path.join(reference_root, "load", "shapes", "human.obj")),
            (MOTION, "make_motion.py", os.path.join(reference_root, "content", "amass_test_17.npz")))
    for out, script, src in jobs:
        if not os.path.exists(out) and os.path.exists(src):
            subprocess.check_call([sys.executable, os.path.join(DIR, script)],
                                  stdout=None if verbose else subprocess.DEVNULL)
