"""Generates tests/golden/*.npz by running the REFERENCE's own code (build container only).

/root/reference/model2_seq.py imports `torchvision` and `mamba_ssm`, neither installed
(ordinary ModuleNotFoundError, nothing denied - SURVEY.md 8c).  This script pre-inserts two stub
modules: `torchvision.models.resnet18/34` returning a small nn.Module ResNet-v1 skeleton written
here (attribute names conv1/bn1/relu/maxpool/layer1-4/avgpool/fc, random init - the pretrained
ImageNet weights need a network fetch), and `mamba_ssm.Mamba` raising if constructed.  It then
imports model2_seq, builds `model2_seq.Encoder(config)` (the GPT variant, model2_seq.py:406-597;
`TransFuser.__init__` :861 wires the Mamba encoder and cannot be constructed without CUDA
mamba_ssm) plus a join MLP of the reference shape (:863-869), loads the oracle's name-hashed state
dict with strict=True (proves names/shapes), runs forward/backward, checks oracle == reference and
stores small golden tensors.  Inputs and weights are regenerated from seeds by
oracle/fusion_ref.py (same torch build on the GPU box => same CPU generator streams), so only
outputs are committed.

Nothing from /root/reference is copied: the fixtures are input seeds + output numbers.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch
from torch import nn

from oracle import fusion_ref as fr
from oracle import train_ref as tr

REF = "/root/reference"


# ------------------------------------------------------------------ stubs --------------------
class _StubBlock(nn.Module):
    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idn)


class _StubResNet(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), layers), start=1):
            blocks = []
            for bi in range(n):
                blocks.append(_StubBlock(inpl, planes, 2 if (li > 1 and bi == 0) else 1))
                inpl = planes
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)


def _install_stubs():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.resnet34 = lambda weights=None, **kw: _StubResNet((3, 4, 6, 3))
    tvm.resnet18 = lambda weights=None, **kw: _StubResNet((2, 2, 2, 2))
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    ms = types.ModuleType("mamba_ssm")

    class Mamba:  # noqa
        def __init__(self, *a, **k):
            raise RuntimeError("mamba_ssm is not available (CUDA-only); GPT variant only")

    ms.Mamba = Mamba
    sys.modules["mamba_ssm"] = ms


def _ref_config(cfg: fr.RefConfig):
    sys.path.insert(0, REF)
    import config_seq  # reference config class (config_seq.py:3-45)

    return config_seq.GlobalConfig(seq_len=cfg.seq_len, n_views=cfg.n_views, add_velocity=cfg.add_velocity,
                                   embd_pdrop=cfg.embd_pdrop, attn_pdrop=cfg.attn_pdrop,
                                   resid_pdrop=cfg.resid_pdrop, n_layer=cfg.n_layer, n_head=cfg.n_head,
                                   block_exp=cfg.block_exp)


def _maxabs(a, b):
    return float((a - b).abs().max())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _install_stubs()
    sys.path.insert(0, REF)
    import model2_seq  # the reference hot-path file

    out = {}
    report = []

    # ---------------- 1. normalize_imagenet (model2_seq.py:36-45) -----------------------------
    x = torch.randint(0, 256, (2, 3, 8, 8), generator=torch.Generator().manual_seed(1)).float()
    report.append(("normalize_imagenet", _maxabs(model2_seq.normalize_imagenet(x), fr.normalize_imagenet(x))))

    # ---------------- 2. GPT stage alone, seq_len=1 (194 tokens), C=64 ------------------------
    cfg1 = fr.RefConfig(seq_len=1, n_layer=2, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    rc1 = _ref_config(cfg1)
    sd1 = fr.make_state(cfg1, seed=7)
    gpt = model2_seq.GPT(n_embd=64, n_head=4, block_exp=4, n_layer=2, vert_anchors=8, horz_anchors=8,
                         seq_len=1, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, config=rc1)
    p = "encoder.transformer1."
    gpt.load_state_dict({k[len(p):]: v for k, v in sd1.items() if k.startswith(p)}, strict=True)
    g = torch.Generator().manual_seed(11)
    B = 2
    img, lid, rad = (torch.randn(B, 64, 8, 8, generator=g) for _ in range(3))
    gps = torch.randn(B, 2, 64, generator=g)
    with torch.no_grad():
        ref_o = gpt(img, lid, rad, gps)
        my_o = fr.gpt_stage(sd1, p, img, lid, rad, gps, cfg1, fr.Ctx(training=True))
    report.append(("gpt_stage(seq_len=1,C=64)", max(_maxabs(a, b) for a, b in zip(ref_o, my_o))))
    out["gpt1_img"] = ref_o[0].numpy()
    out["gpt1_lid"] = ref_o[1].numpy()
    out["gpt1_rad"] = ref_o[2].numpy()
    out["gpt1_gps"] = ref_o[3].numpy()

    # ---------------- 3. full Encoder + join, seq_len=5, B=2, train-mode BN, dropout 0 ---------
    cfg = fr.RefConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    rc = _ref_config(cfg)
    sd = fr.make_state(cfg, seed=3)
    enc = model2_seq.Encoder(rc)
    join = nn.Sequential(nn.Linear(512, 256), nn.ReLU(inplace=True), nn.Linear(256, 128),
                         nn.ReLU(inplace=True), nn.Linear(128, 64))  # shape of model2_seq.py:863-869
    enc.load_state_dict({k[len("encoder."):]: v.clone() for k, v in sd.items() if k.startswith("encoder.")},
                        strict=True)
    join.load_state_dict({k[len("join."):]: v.clone() for k, v in sd.items() if k.startswith("join.")},
                         strict=True)
    n_params = sum(p.numel() for p in enc.parameters()) + sum(p.numel() for p in join.parameters())
    assert n_params == 78_422_528, n_params
    B = 2
    imgs, lids, rads, gps, target, beamidx = fr.make_inputs(cfg, B, seed=100)
    enc.train()
    join.train()
    fused_ref = enc(imgs, lids, rads, gps)
    logits_ref = join(fused_ref)
    loss_ref = tr.sigmoid_focal_loss(logits_ref, target)
    loss_ref.backward()

    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    cap = {}
    logits_my = fr.transfuser_forward(sdo, imgs, lids, rads, gps, cfg, fr.Ctx(training=True, capture=cap))
    loss_my = tr.sigmoid_focal_loss(logits_my, target)
    loss_my.backward()
    report.append(("encoder fused (B=2)", _maxabs(fused_ref, cap["fused"])))
    report.append(("logits (B=2)", _maxabs(logits_ref, logits_my)))
    report.append(("loss", abs(float(loss_ref) - float(loss_my))))
    ref_named = dict(("encoder." + k, v) for k, v in enc.named_parameters())
    ref_named.update(("join." + k, v) for k, v in join.named_parameters())
    worst = 0.0
    for k, v in ref_named.items():
        d = _maxabs(v.grad, sdo[k].grad) / (float(v.grad.abs().max()) + 1e-12)
        worst = max(worst, d)
    report.append(("grads (max rel-to-max over all params)", worst))
    ref_bufs = dict(("encoder." + k, v) for k, v in enc.named_buffers())
    worst = max(_maxabs(v.float(), sdo[k].float()) for k, v in ref_bufs.items())
    report.append(("BN running stats after 1 step", worst))

    out["logits_b2"] = logits_ref.detach().numpy()
    out["fused_b2"] = fused_ref.detach().numpy()
    out["loss_b2"] = np.array(float(loss_ref), dtype=np.float64)
    for name in ("encoder.image_encoder.features.conv1.weight", "encoder.transformer4.blocks.7.mlp.2.weight",
                 "encoder.transformer1.pos_emb", "encoder.vel_emb1.weight", "join.4.bias",
                 "encoder.radar_encoder._model.layer4.1.bn2.weight"):
        gname = "grad:" + name
        gt = ref_named[name].grad
        out[gname + ":absmax"] = np.array(float(gt.abs().max()))
        out[gname + ":l2"] = np.array(float(gt.norm()))
        out[gname + ":head"] = gt.flatten()[:16].numpy()
    out["bn:encoder.image_encoder.features.bn1.running_mean"] = ref_bufs[
        "encoder.image_encoder.features.bn1.running_mean"].numpy()
    out["bn:encoder.image_encoder.features.bn1.running_var"] = ref_bufs[
        "encoder.image_encoder.features.bn1.running_var"].numpy()

    # eval-mode forward (running stats) on the updated buffers
    enc.eval(); join.eval()
    with torch.no_grad():
        logits_eval_ref = join(enc(imgs, lids, rads, gps))
        logits_eval_my = fr.transfuser_forward(sdo, imgs, lids, rads, gps, cfg, fr.Ctx(training=False))
    report.append(("logits eval-mode", _maxabs(logits_eval_ref, logits_eval_my)))
    out["logits_eval_b2"] = logits_eval_ref.numpy()

    print("oracle vs reference (max abs diff):")
    bad = False
    for k, v in report:
        print(f"  {k:45s} {v:.3e}")
        if not (v < 5e-5):
            bad = True
    np.savez(os.path.join(HERE, "fusion_golden.npz"), **out)
    with open(os.path.join(HERE, "oracle_vs_reference.txt"), "w") as f:
        f.write("max abs diff, oracle/fusion_ref.py vs /root/reference/model2_seq.py (torch %s, CPU)\n" % torch.__version__)
        for k, v in report:
            f.write(f"{k:45s} {v:.3e}\n")
    if bad:
        raise SystemExit("oracle does not match reference")


if __name__ == "__main__":
    main()
