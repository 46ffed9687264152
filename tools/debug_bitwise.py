"""Ad-hoc: per-field mismatch report of HIP preprocess records vs the fp32 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle
from abi_runner import RawCall
from helpers import make_scene, oracle_settings
sc = make_scene(P=2000, sh_degree=3, seed=3, H=128, W=96, spread=0.6)
rc = RawCall(sc); assert rc.forward() == 0
rec = rc.geom_records()
pre = oracle.preprocess(sc["means3D"], None, sc["shs"], None, sc["opacities"], sc["scales"], sc["rotations"], None, oracle_settings(sc))
vis = pre["visible"].numpy()
for name, ref in (("mx", pre["mean2D"][:, 0]), ("my", pre["mean2D"][:, 1]), ("ca", pre["conic"][:, 0]), ("cb", pre["conic"][:, 1]), ("cc", pre["conic"][:, 2]), ("depth", pre["depth"]), ("r", pre["rgb"][:,0]), ("g", pre["rgb"][:,1]), ("b", pre["rgb"][:,2])):
    a, b = rec[name][vis], ref.numpy()[vis]
    bad = a.view(np.uint32) != b.view(np.uint32)
    print(name, int(bad.sum()), "of", len(a), "max rel", float(np.abs((a-b)/np.maximum(np.abs(b),1e-30)).max()))
