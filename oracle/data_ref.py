"""CPU oracle (TEST INFRASTRUCTURE ONLY - imported by tests/, never by the product path) for the per-sample input
work of CARLA_Data.__getitem__, /root/reference/data2_seq.py:42-173: numpy restatements of

* `lidar_to_histogram_features` (data2_seq.py:177-211): 256x256 BEV occupancy histogram over np.linspace bin edges
  via np.histogramdd, clipped at 5 points per cell, scaled by 1/5, shape (1, 256, 256) float64;
* the per-scenario custom field of view (data2_seq.py:190-202);
* the Gaussian soft beam target (data2_seq.py:160-170);
* the image HWC uint8 -> CHW tensor step (data2_seq.py:110-146) followed by the float32 cast of the training loop
  (train2_seq.py:111-116) and `normalize_imagenet` (model2_seq.py:36-45).

Pinned: `lidar_bev` is checked against the reference's own function (imported with stubs for the absent
open3d / utm / cv2 / torchvision modules) by tests/golden/make_golden_data.py -> tests/golden/data_golden.npz.
The soft target, the flip augmentation of every modality and the frame / radar tensor layout live inline in `__getitem__`:
`getitem()` below restates them, and tests/golden/make_golden_getitem.py runs the reference's own `__getitem__` on the same
synthetic in-memory files (its three file readers pointed at a table) -> tests/golden/getitem_golden.npz, max abs diff 0.0.
"""
from __future__ import annotations

import numpy as np

HIST_MAX_PER_PIXEL = 5  # data2_seq.py:184
NBINS = 256


def fov_edges(address: str = "", custom_fov: bool = False):
    """-> (xbins, ybins): the 257 bin edges per axis, data2_seq.py:185-202."""
    xbins = np.linspace(-50, 0, 257)
    ybins = np.linspace(-50, 50, 257)
    if custom_fov:
        if "scenario31" in address:
            xbins, ybins = np.linspace(-70, 0, 257), np.linspace(-25, 14, 257)
        elif "scenario32" in address:
            xbins, ybins = np.linspace(-60, 0, 257), np.linspace(-40, 5.5, 257)
        elif "scenario33" in address:
            xbins, ybins = np.linspace(-50, 0, 257), np.linspace(-12, 7, 257)
        elif "scenario34" in address:
            xbins, ybins = np.linspace(-50, 0, 257), np.linspace(-20, 10, 257)
    return xbins, ybins


def lidar_bev(points, address: str = "", custom_fov: bool = False, flip: bool = False):
    """points (P, >=2) float64 -> (1, 256, 256) float64, data2_seq.py:177-211 (+ the flip at :157-158)."""
    xbins, ybins = fov_edges(address, custom_fov)
    hist = np.histogramdd(np.asarray(points)[..., :2], bins=(xbins, ybins))[0]
    hist[hist > HIST_MAX_PER_PIXEL] = HIST_MAX_PER_PIXEL
    out = (hist / HIST_MAX_PER_PIXEL)[np.newaxis, :, :]
    if flip:
        out = np.ascontiguousarray(np.flip(out, 2))
    return out


def lidar_counts(points, xbins, ybins):
    """raw integer cell counts (before the clip), for bit-exact checks of the scatter kernel"""
    return np.histogramdd(np.asarray(points)[..., :2], bins=(xbins, ybins))[0].astype(np.int64)


def soft_beam_target(beamidx: int, flip: bool = False):
    """-> (beam (64,) float64, beamidx) of data2_seq.py:160-170; beamidx is 0-based here (the csv value - 1)."""
    from scipy import stats
    x_data = range(max(beamidx - 5, 0), min(beamidx + 5, 63) + 1)
    y_data = stats.norm.pdf(x_data, beamidx, 0.5)
    data_beam = np.zeros((64))
    data_beam[x_data] = y_data * 1.25
    if flip:
        beamidx = 63 - beamidx
        data_beam = np.ascontiguousarray(np.flip(data_beam, 0))
    return data_beam, beamidx


def image_to_input(img_hwc_u8, flip: bool = False):
    """decoded (256, 256, 3) uint8 frame -> normalised (3, 256, 256) float32 exactly as the reference path computes it:
    flip (data2_seq.py:144-146), HWC->CHW (:147), float32 cast (train2_seq.py:111), normalize_imagenet
    (model2_seq.py:36-45) with float32 arithmetic."""
    import torch
    imgs = np.asarray(img_hwc_u8)
    if flip:
        imgs = np.ascontiguousarray(np.flip(imgs, 1))
    x = torch.from_numpy(np.transpose(imgs, (2, 0, 1)).copy()).to(torch.float32)[None]
    x = x.clone()
    x[:, 0] = ((x[:, 0] / 255.0) - 0.485) / 0.229
    x[:, 1] = ((x[:, 1] / 255.0) - 0.456) / 0.224
    x[:, 2] = ((x[:, 2] / 255.0) - 0.406) / 0.225
    return x[0].numpy()


def make_cloud(n: int, seed: int, xbins=None, ybins=None):
    """synthetic LiDAR cloud (n, 3) float64 that exercises the histogram's edge cases: clustered cells (> 5 hits),
    points exactly on interior edges and on the first / last edge, outliers on every side, and a NaN."""
    if xbins is None:
        xbins, ybins = fov_edges()
    rng = np.random.default_rng(seed)
    pts = np.empty((n, 3))
    pts[:, 0] = rng.uniform(xbins[0] - 5, xbins[-1] + 5, n)
    pts[:, 1] = rng.uniform(ybins[0] - 5, ybins[-1] + 5, n)
    pts[:, 2] = rng.uniform(-2, 4, n)
    if n < 8:  # too small for the special points below (an empty cloud is a legal input)
        return pts
    k = n // 10
    pts[:k, 0] = rng.normal(-20.0, 0.15, k)      # dense cluster: cells above the clip
    pts[:k, 1] = rng.normal(3.0, 0.15, k)
    e = min(64, n // 20)
    pts[k:k + e, 0] = rng.choice(xbins, e)        # exactly on edges (incl. possibly first / last)
    pts[k + e:k + 2 * e, 1] = rng.choice(ybins, e)
    pts[k + 2 * e] = (xbins[-1], ybins[-1], 0.0)  # last edge of both axes: belongs to the last bin
    pts[k + 2 * e + 1] = (xbins[0], ybins[0], 0.0)
    pts[k + 2 * e + 2] = (np.nextafter(xbins[-1], np.inf), 0.0, 0.0)  # just outside
    pts[k + 2 * e + 3] = (np.nan, 0.0, 0.0)
    return pts


# ----------------------------------------------------------------------------------------------------------------------
# one whole sample, as CARLA_Data.__getitem__ (data2_seq.py:42-173) builds it from decoded files
def make_getitem_files(seed: int, scen: int, beam1: int, custom_fov: bool):
    """synthetic decoded "files" of one 5-frame sample -> (files: path -> array, dataframe columns, gps (2, 2)).
    Paths follow the reference's rewriting rules for augment = {camera: 1, lidar: 0, radar: 0} (data2_seq.py:64-87)."""
    rng = np.random.default_rng(seed)
    base = f"scenario{scen}/unit1/"
    files, frame = {}, {"unit1_beam": [beam1]}
    for k in range(1, 6):
        frame[f"unit1_rgb_{k}"] = [base + f"camera_data/image_{k}.jpg"]
        frame[f"unit1_lidar_{k}"] = [base + f"lidar_data/cloud_{k}.ply"]
        frame[f"unit1_radar_{k}"] = [base + f"radar_data/radar_{k}.npy"]
        files[base + f"camera_data_aug/image_{k}_1.jpg"] = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
        xb, yb = fov_edges(base, custom_fov)
        files[base + f"lidar_data/cloud_{k}.ply"] = make_cloud(4000 + 100 * k, seed * 10 + k, xb, yb)
        files[base + f"radar_data_ang/radar_{k}.npy"] = rng.random((256, 256), dtype=np.float32)
        files[base + f"radar_data_vel/radar_{k}.npy"] = rng.standard_normal((256, 256), dtype=np.float32)
    gps = rng.uniform(-1, 1, (2, 2))
    return files, frame, gps


def getitem(seed: int, scen: int, beam1: int, flip: bool, custom_fov: bool, add_velocity: int):
    """the oracle's restatement of one sample: fronts 5 x (3, 256, 256) uint8 (HWC -> CHW view, :147), radars 5 x (1 | 2,
    256, 256) float32 (angle [+ velocity] maps stacked, :148-154), lidars 5 x (1, 256, 256) float64 (:155-159), gps (2, 2)
    with the second coordinate negated under flip (:49-50), the Gaussian soft beam target and the beam index (:160-171)."""
    files, frame, gps = make_getitem_files(seed, scen, beam1, custom_fov)
    base = f"scenario{scen}/unit1/"
    out = dict(fronts=[], radars=[], lidars=[])
    g = gps.copy()
    if flip:
        g[:, 1] = -g[:, 1]
    out["gps"] = g
    for k in range(1, 6):
        img = files[base + f"camera_data_aug/image_{k}_1.jpg"]
        ang = files[base + f"radar_data_ang/radar_{k}.npy"]
        vel = files[base + f"radar_data_vel/radar_{k}.npy"]
        if flip:
            img, ang, vel = (np.ascontiguousarray(np.flip(a, 1)) for a in (img, ang, vel))
        out["fronts"].append(np.transpose(img, (2, 0, 1)))
        out["radars"].append(np.concatenate([ang[None], vel[None]], 0) if add_velocity else ang[None])
        path = base + f"lidar_data/cloud_{k}.ply"
        out["lidars"].append(lidar_bev(files[path], path, custom_fov, flip))
    out["beam"], out["beamidx"] = soft_beam_target(beam1 - 1, flip)
    return out
