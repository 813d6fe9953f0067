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
The soft target and the image step live inline in `__getitem__` (needs the dataset on disk) and are restated from
the source text; they are pinned by the closed forms they spell out (scipy.stats.norm.pdf, np.transpose).
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
