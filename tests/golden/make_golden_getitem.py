"""Generates tests/golden/getitem_golden.npz by running the REFERENCE's own `CARLA_Data.__getitem__`
(/root/reference/data2_seq.py:42-173) end to end on synthetic in-memory "files" (build container only).

`data2_seq.py` imports with empty stub modules for the absent open3d / utm / cv2 / torchvision (as make_golden_data.py).
`CARLA_Data.__init__` reads a csv and normalises GPS with `utm`, so the dataset object is created with `object.__new__`
and given exactly the attributes `__getitem__` reads (dataframe columns, pos_input_normalized, flags).  The three file
readers it calls are pointed at an in-memory table: `Image.open(path).resize((256, 256))`, `np.load(path)` and
`o3d.io.read_point_cloud(path).points` return seeded synthetic arrays keyed by path (the module's `np` is a proxy that
forwards everything but `load`).  Every other line - the path rewriting, the scenario lookup, the flip augmentation of
frames / radar maps / BEV / GPS / beam (:49-50,144-146,151-152,157-158,168-170), the HWC->CHW transpose, the radar channel
stacking, `lidar_to_histogram_features`, and the Gaussian soft beam target (:160-167) - is the reference's, executed as is.

Stored per case: the seeds that rebuild the synthetic files (oracle/data_ref.py::make_getitem_files), the full beam / beamidx /
gps outputs, and strided samples + sums of fronts / radars / lidars (the full tensors are compared on the spot: the oracle
must equal the reference with max abs diff 0.0, recorded in oracle_vs_reference.txt).
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_getitem.py
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

FILES = {}


class _NpProxy:
    """the reference module's `np`: numpy itself, except that load() serves the in-memory table"""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def load(path, *a, **k):
        return FILES[path].copy()


class _Img:
    def __init__(self, arr):
        self.arr = arr

    def resize(self, size):
        assert size == (256, 256)
        return self

    def __array__(self, dtype=None, copy=None):
        return self.arr.copy()


ref.np = _NpProxy()
ref.Image = types.SimpleNamespace(open=lambda path: _Img(FILES[path]))
ref.o3d = types.SimpleNamespace(io=types.SimpleNamespace(read_point_cloud=lambda path: types.SimpleNamespace(points=FILES[path])))

# (seed, scenario, 1-based beam of the csv, flip, custom_FoV, add_velocity)
CASES = [(11, 31, 1, False, False, 1), (12, 32, 64, True, True, 1), (13, 33, 4, True, False, 1), (14, 34, 33, False, True, 0),
         (15, 31, 61, True, True, 1), (16, 32, 6, False, True, 1)]


def run_reference(seed, scen, beam1, flip, fov, vel):
    files, frame, gps = dr.make_getitem_files(seed, scen, beam1, fov)
    FILES.clear()
    FILES.update(files)
    ds = object.__new__(ref.CARLA_Data)
    ds.dataframe = frame
    ds.root = ""
    ds.pos_input_normalized = gps.copy()[None]
    ds.test = False
    ds.add_velocity = vel
    ds.add_mask = False
    ds.enhanced = False
    ds.filtered = False
    ds.augment = {"camera": 1, "lidar": 0, "radar": 0}   # camera > 0: the plain Image.open(...).resize branch (:139-141)
    ds.custom_FoV_lidar = fov
    ds.flip = flip
    ds.add_seg = False
    return ds.__getitem__(0)


def main():
    out = {}
    worst = 0.0
    for case in CASES:
        seed, scen, beam1, flip, fov, vel = case
        got = run_reference(*case)
        mine = dr.getitem(seed, scen, beam1, flip, fov, vel)
        assert got["scenario"] == f"scenario{scen}" and got["loss_weight"] == 1.0
        key = f"case{seed}"
        out[key + "_meta"] = np.array([seed, scen, beam1, int(flip), int(fov), vel])
        pairs = [("beam", np.asarray(got["beam"][0]), mine["beam"]), ("gps", np.asarray(got["gps"]), mine["gps"]),
                 ("beamidx", np.asarray(got["beamidx"][0]), np.asarray(mine["beamidx"]))]
        for t in range(5):
            pairs.append((f"front{t}", got["fronts"][t].numpy(), mine["fronts"][t]))
            pairs.append((f"radar{t}", got["radars"][t].numpy(), mine["radars"][t]))
            pairs.append((f"lidar{t}", np.asarray(got["lidars"][t]), mine["lidars"][t]))
        for name, a, b in pairs:
            assert a.shape == b.shape and a.dtype == b.dtype, (name, a.shape, b.shape, a.dtype, b.dtype)
            worst = max(worst, float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()))
            if name in ("beam", "gps", "beamidx"):
                out[f"{key}_{name}"] = a
            else:
                flat = a.reshape(-1)
                out[f"{key}_{name}_sample"] = flat[::997].copy()
                out[f"{key}_{name}_sum"] = np.array(flat.astype(np.float64).sum())
    np.savez_compressed(os.path.join(HERE, "getitem_golden.npz"), **out)
    log = os.path.join(HERE, "oracle_vs_reference.txt")
    kept = [l for l in open(log).read().splitlines() if not l.startswith("data_ref.getitem")] if os.path.exists(log) else []
    kept.append(f"data_ref.getitem vs data2_seq.CARLA_Data.__getitem__ (frames, radar, BEV, gps, soft beam target, flip), "
                f"{len(CASES)} samples: max abs diff {worst}")
    open(log, "w").write("\n".join(kept) + "\n")
    print("max abs diff oracle vs reference:", worst, "cases:", len(CASES))
    if worst != 0.0:
        raise SystemExit("oracle does not match reference")


if __name__ == "__main__":
    main()
