"""Generates tests/golden/data_golden.npz by running the REFERENCE's own `lidar_to_histogram_features`
(/root/reference/data2_seq.py:177-211) on seeded synthetic clouds (build container only).

data2_seq.py imports open3d, utm, cv2 and torchvision, none installed (ordinary ModuleNotFoundError); they are only
used by the file/dataset code, not by the histogram function, so empty stub modules are pre-inserted.  Nothing from
/root/reference is copied: the fixture is (seed, cloud size, scenario) inputs + the reference's output histograms,
stored sparsely.  Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_data.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np

from oracle import data_ref as dr

for name in ("open3d", "utm", "cv2", "torchvision", "torchvision.transforms"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.path.insert(0, "/root/reference")
import data2_seq as ref  # noqa: E402

CASES = [  # (seed, n points, address, custom_FoV)
    (1, 20000, "scenario31/x.ply", False),
    (2, 20000, "scenario31/x.ply", True),
    (3, 5000, "scenario32/x.ply", True),
    (4, 3000, "scenario33/x.ply", True),
    (5, 12345, "scenario34/x.ply", True),
    (6, 16, "scenario34/x.ply", False),
]
out = {}
maxdiff = 0.0
for seed, n, addr, fov in CASES:
    xb, yb = dr.fov_edges(addr, fov)
    pts = dr.make_cloud(n, seed, xb, yb)
    want = ref.lidar_to_histogram_features(pts, addr, custom_FoV=fov)
    got = dr.lidar_bev(pts, addr, fov)
    maxdiff = max(maxdiff, float(np.abs(want - got).max()))
    assert want.shape == (1, 256, 256) and want.dtype == np.float64
    nz = np.nonzero(want[0])
    out[f"case{seed}_meta"] = np.array([seed, n, int(fov), int(addr[8:10])])
    out[f"case{seed}_idx"] = np.stack(nz).astype(np.int16)
    out[f"case{seed}_val"] = want[0][nz]
np.savez_compressed(os.path.join(HERE, "data_golden.npz"), **out)
log = os.path.join(HERE, "oracle_vs_reference.txt")
kept = [l for l in open(log).read().splitlines() if not l.startswith("data_ref.lidar_bev")] if os.path.exists(log) else []
kept.append(f"data_ref.lidar_bev vs data2_seq.lidar_to_histogram_features, {len(CASES)} clouds: max abs diff {maxdiff}")
open(log, "w").write("\n".join(kept) + "\n")
print("max abs diff oracle vs reference:", maxdiff, "cases:", len(CASES))
