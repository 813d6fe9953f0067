"""Synthetic DeepSense6G-shaped batches (there is no dataset offline).

Value ranges follow the reference loader's output dict (/root/reference/data2_seq.py): images are
uint8-valued fp32 (:141), LiDAR BEV histograms in {0,.2,..,1} and mostly empty (:204-211), radar
range-angle / range-velocity maps min-max scaled to [0,1] (Data_Preprocessing/
Radar_data_preprocessing.py:22-23,41-42), GPS = normalised angle in radians duplicated in both
columns (:273-280), target = 1.25 * N(k; beam, 0.5) on beam-5..beam+5 (:162-167).
`learnable=True` paints the beam index into every modality so that DBA on a held-out seed range is
meaningful.
"""
from __future__ import annotations

import math

import torch


def soft_beam_target(beamidx: torch.Tensor) -> torch.Tensor:
    k = torch.arange(64, dtype=torch.float64)[None, :]
    mu = beamidx.to(torch.float64)[:, None]
    pdf = torch.exp(-0.5 * ((k - mu) / 0.5) ** 2) / (0.5 * math.sqrt(2 * math.pi))
    lo, hi = (mu - 5).clamp(min=0), (mu + 5).clamp(max=63)
    return (1.25 * pdf * ((k >= lo) & (k <= hi))).float()


def make_batch(batch, seq_len=5, n_views=1, add_velocity=1, seed=100, device="cpu", learnable=False, res=256):
    """-> (fronts, lidars, radars, gps, soft_target, beamidx) with the reference's shapes:
    seq_len*n_views x (B,3,R,R), seq_len x (B,1,R,R), seq_len x (B,2,R,R), (B,2,2), (B,64), (B,)."""
    g = torch.Generator().manual_seed(seed)
    rc = 2 if add_velocity else 1
    beam = torch.randint(0, 64, (batch,), generator=g)
    fronts = [torch.randint(0, 256, (batch, 3, res, res), generator=g).float() for _ in range(seq_len * n_views)]
    lidars, radars = [], []
    for _ in range(seq_len):
        occ = (torch.rand(batch, 1, res, res, generator=g) < 0.05).float()
        lidars.append(occ * torch.randint(1, 6, (batch, 1, res, res), generator=g).float() / 5.0)
        radars.append(torch.rand(batch, rc, res, res, generator=g))
    if learnable:
        w = res // 64
        for b in range(batch):
            c0 = int(beam[b]) * w
            for t in fronts:
                t[b, :, :, c0:c0 + w] = 255.0
            for t in lidars + radars:
                t[b, :, :, c0:c0 + w] = 1.0
        ang = (beam.float() / 63.0 - 0.5) * math.pi
        ang = ang[:, None].repeat(1, 2) + 0.01 * torch.randn(batch, 2, generator=g)
    else:
        ang = (torch.rand(batch, 2, generator=g) - 0.5) * math.pi
    gps = ang[:, :, None].repeat(1, 1, 2).contiguous()
    tgt = soft_beam_target(beam)
    mv = lambda t: t.to(device)
    return [mv(t) for t in fronts], [mv(t) for t in lidars], [mv(t) for t in radars], mv(gps), mv(tgt), beam
