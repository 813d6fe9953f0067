"""CPU: the input-pipeline oracle (oracle/data_ref.py) against the reference's own outputs (tests/golden/data_golden.npz,
made by tests/golden/make_golden_data.py from data2_seq.lidar_to_histogram_features) and against the closed forms the
reference spells out for the soft target."""
import os

import numpy as np

from oracle import data_ref as dr

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data_golden.npz")


def test_lidar_bev_matches_reference_golden():
    g = np.load(GOLD)
    seeds = sorted(int(k[4:-5]) for k in g.files if k.endswith("_meta"))
    assert len(seeds) == 6
    for sd in seeds:
        seed, n, fov, scen = (int(v) for v in g[f"case{sd}_meta"])
        addr = f"scenario{scen}/x.ply"
        xb, yb = dr.fov_edges(addr, bool(fov))
        pts = dr.make_cloud(n, seed, xb, yb)
        want = np.zeros((256, 256))
        idx = g[f"case{sd}_idx"].astype(np.int64)
        want[idx[0], idx[1]] = g[f"case{sd}_val"]
        got = dr.lidar_bev(pts, addr, bool(fov))
        assert got.shape == (1, 256, 256) and got.dtype == np.float64
        assert np.array_equal(got[0], want), sd
        assert set(np.unique(want)) <= {0.0, 0.2, 0.4, 0.6, 0.8, 1.0} and (n < 1000 or want.max() == 1.0)


def test_histogram_edge_semantics():
    xb, yb = dr.fov_edges()
    pts = np.array([[xb[0], yb[0], 0], [xb[-1], yb[-1], 0], [xb[5], yb[7], 0], [np.nextafter(xb[-1], 1), 0, 0],
                    [np.nextafter(xb[0], -100), 0, 0], [np.nan, 0, 0], [-10.0, np.inf, 0]])
    c = dr.lidar_counts(pts, xb, yb)
    assert c.sum() == 3 and c[0, 0] == 1 and c[255, 255] == 1 and c[5, 7] == 1


def test_soft_beam_target_closed_form():
    for idx in (0, 3, 31, 60, 63):
        beam, bi = dr.soft_beam_target(idx)
        assert bi == idx and beam.shape == (64,)
        for k in range(64):
            want = 1.25 * np.exp(-((k - idx) / 0.5) ** 2 / 2) / (0.5 * np.sqrt(2 * np.pi)) if abs(k - idx) <= 5 else 0.0
            assert abs(beam[k] - want) <= 1e-15 * max(1.0, want)
        fb, fi = dr.soft_beam_target(idx, flip=True)
        assert fi == 63 - idx and np.array_equal(fb, beam[::-1])
    # the package's own host-side generator (synthetic batches) agrees with the restated reference formula
    from deepsense6g_tii_amd.synthetic import soft_beam_target as pkg_target
    import torch
    got = pkg_target(torch.tensor([0, 31, 63]))
    want = np.stack([dr.soft_beam_target(i)[0] for i in (0, 31, 63)]).astype(np.float32)
    assert np.allclose(got.numpy(), want, rtol=1e-6, atol=1e-12)


GOLD_ITEM = os.path.join(os.path.dirname(__file__), "golden", "getitem_golden.npz")


def getitem_cases():
    g = np.load(GOLD_ITEM)
    return g, [tuple(int(v) for v in g[k]) for k in sorted(g.files) if k.endswith("_meta")]


def test_whole_sample_matches_reference_getitem_fixture():
    """oracle/data_ref.py::getitem against the outputs of the reference's own CARLA_Data.__getitem__ run end to end on the same
    synthetic files (tests/golden/make_golden_getitem.py): soft beam target, beam index and GPS in full, frames / radar maps /
    BEV histograms by strided samples and sums - flip on and off, custom field of view, with and without the velocity map,
    beams at both ends of the codebook."""
    g, cases = getitem_cases()
    assert len(cases) == 6 and {c[3] for c in cases} == {0, 1} and {c[2] for c in cases} >= {1, 64}
    for seed, scen, beam1, flip, fov, vel in cases:
        mine = dr.getitem(seed, scen, beam1, bool(flip), bool(fov), vel)
        key = f"case{seed}"
        assert np.array_equal(mine["beam"], g[key + "_beam"]) and int(mine["beamidx"]) == int(g[key + "_beamidx"])
        assert np.array_equal(mine["gps"], g[key + "_gps"])
        for t in range(5):
            for name, arr in ((f"front{t}", mine["fronts"][t]), (f"radar{t}", mine["radars"][t]), (f"lidar{t}", mine["lidars"][t])):
                flat = np.ascontiguousarray(arr).reshape(-1)
                want = g[f"{key}_{name}_sample"]
                assert flat.dtype == want.dtype and np.array_equal(flat[::997], want), (key, name)
                assert float(flat.astype(np.float64).sum()) == float(g[f"{key}_{name}_sum"]), (key, name)
        assert mine["radars"][0].shape == ((2 if vel else 1), 256, 256)
