"""Generates humangaussian_amd/data/amass_test_17_poses.npz - the 136 SMPL-X pose frames of the reference's demo motion.
A LOCAL, git-ignored artefact (the clip is AMASS data, whose licence forbids redistribution: it is not shipped with this
repository); `__graft_entry__.build()` runs this script wherever /root/reference exists.  Without the file
`animation.MotionDriver` plays its procedural sway of the same nine joints and period, and says so.  A user's own
AMASS-format clip (`poses` (F, 55, 3)) can be passed as `MotionDriver(poses_path=...)`.

  python humangaussian_amd/data/make_motion.py

Source: /root/reference/content/amass_test_17.npz (`poses` (136, 55, 3) axis-angle per SMPL-X joint, `trans` (136, 3)):
the sequence `animation.py --motion content/amass_test_17.npz --play` plays (animation.py:311-330 reads `poses[i]`,
:966-1004 loops over the frames).  Stored as float32.  The SMPL-X model files that turn poses into vertices are
not in the reference tree; humangaussian_amd/animation.py::MotionDriver drives a toy articulation of the committed
human mesh with these angles instead (BASELINE.json configs[4]).
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/content/amass_test_17.npz"

if __name__ == "__main__":
    d = np.load(SRC, allow_pickle=True)
    out = os.path.join(HERE, "amass_test_17_poses.npz")
    np.savez_compressed(out, poses=d["poses"].astype(np.float32), trans=d["trans"].astype(np.float32),
                        mocap_framerate=np.int32(d["mocap_framerate"]), source="content/amass_test_17.npz")
    print(out, d["poses"].shape, os.path.getsize(out))
