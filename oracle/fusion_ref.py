"""CPU ORACLE (test infrastructure, never the product path).

Plain-torch fp32 restatement of the reference hot path: the TransFuser-style multimodal fusion
forward of /root/reference/model2_seq.py (class ``Encoder`` :406-597 with ``GPT`` :175-287,
``Block`` :113-134, ``SelfAttention`` :74-110, ``normalize_imagenet`` :36-45) followed by the
``join`` MLP (:863-869, :890).  torchvision is not installed here, so the ResNet-18/34 trunks
(reference call sites :23, :59, :495-512, :528-530, :546-548, :565-567, :581-587) are restated from
the published ResNet-v1 BasicBlock definition: stem 7x7/2 p3 -> BN -> ReLU -> maxpool 3x3/2 p1;
BasicBlock = conv3x3(s) -> BN -> ReLU -> conv3x3 -> BN -> (+identity | conv1x1(s) -> BN) -> ReLU;
no conv bias; BN eps 1e-5, momentum 0.1.

Written *functionally* over a flat name->tensor dict whose keys/shapes are exactly the reference
state-dict, so the same dict loads into the reference ``Encoder`` with
``strict=True`` (done by tests/golden/make_golden.py in the build container, which is how this
file is pinned: see tests/golden/README.md).

Parity status: pinned against the reference's own code imported in the build container
(model2_seq.Encoder + join of the reference shape) to <=2e-6 abs on logits; the torchvision
boundary (ResNet trunk, sigmoid_focal_loss) is restated from the published definitions and is
"parity unpinned" at that boundary (the reference holds no fixtures for it).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict
from dataclasses import dataclass

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# configuration: shape contract of /root/reference/config_seq.py:3-45
# ----------------------------------------------------------------------------------------------
@dataclass
class RefConfig:
    seq_len: int = 5
    n_views: int = 1
    vert_anchors: int = 8
    horz_anchors: int = 8
    n_head: int = 4
    block_exp: int = 4
    n_layer: int = 8
    embd_pdrop: float = 0.1
    attn_pdrop: float = 0.1
    resid_pdrop: float = 0.1
    add_velocity: int = 1
    pred_len: int = 4
    gru_head: bool = False  # the 30->5 variant's GRU beam-sequence head (model2_seq_30to5.py:842-862)

    @property
    def n_tokens(self) -> int:  # model2_seq.py:189
        return (self.n_views + 2) * self.seq_len * self.vert_anchors * self.horz_anchors + 2


STAGE_WIDTH = (64, 128, 256, 512)
RESNET_LAYERS = {"resnet34": (3, 4, 6, 3), "resnet18": (2, 2, 2, 2)}
IMAGENET_MEAN = (0.485, 0.456, 0.406)  # model2_seq.py:42-44
IMAGENET_STD = (0.229, 0.224, 0.225)


def trunk_prefixes():
    """(prefix, arch, in_channels-if-replaced) of the three trunks, model2_seq.py:415-420."""
    return (
        ("encoder.image_encoder.features.", "resnet34"),
        ("encoder.lidar_encoder._model.", "resnet18"),
        ("encoder.radar_encoder._model.", "resnet18"),
    )


# ----------------------------------------------------------------------------------------------
# parameter table: names and shapes
# ----------------------------------------------------------------------------------------------
def param_shapes(cfg: RefConfig) -> "OrderedDict[str, tuple]":
    """Ordered name -> shape of every parameter *and* BN buffer, in reference state-dict order."""
    out: "OrderedDict[str, tuple]" = OrderedDict()

    def bn(prefix, c):
        out[prefix + "weight"] = (c,)
        out[prefix + "bias"] = (c,)
        out[prefix + "running_mean"] = (c,)
        out[prefix + "running_var"] = (c,)
        out[prefix + "num_batches_tracked"] = ()

    def trunk(prefix, arch, cin):
        out[prefix + "conv1.weight"] = (64, cin, 7, 7)
        bn(prefix + "bn1.", 64)
        inplanes = 64
        for li, (planes, nblk) in enumerate(zip(STAGE_WIDTH, RESNET_LAYERS[arch]), start=1):
            for bi in range(nblk):
                p = f"{prefix}layer{li}.{bi}."
                stride = 2 if (li > 1 and bi == 0) else 1
                out[p + "conv1.weight"] = (planes, inplanes, 3, 3)
                bn(p + "bn1.", planes)
                out[p + "conv2.weight"] = (planes, planes, 3, 3)
                bn(p + "bn2.", planes)
                if stride != 1 or inplanes != planes:
                    out[p + "downsample.0.weight"] = (planes, inplanes, 1, 1)
                    bn(p + "downsample.1.", planes)
                inplanes = planes

    radar_c = 2 if cfg.add_velocity else 1
    for (prefix, arch), cin in zip(trunk_prefixes(), (3, 1, radar_c)):
        trunk(prefix, arch, cin)

    widths = (2,) + STAGE_WIDTH
    for s in range(1, 5):
        out[f"encoder.vel_emb{s}.weight"] = (widths[s], widths[s - 1])
        out[f"encoder.vel_emb{s}.bias"] = (widths[s],)

    for s, c in enumerate(STAGE_WIDTH, start=1):
        p = f"encoder.transformer{s}."
        out[p + "pos_emb"] = (1, cfg.n_tokens, c)
        for i in range(cfg.n_layer):
            b = f"{p}blocks.{i}."
            for ln in ("ln1", "ln2"):
                out[b + ln + ".weight"] = (c,)
                out[b + ln + ".bias"] = (c,)
            for lin in ("key", "query", "value", "proj"):
                out[b + f"attn.{lin}.weight"] = (c, c)
                out[b + f"attn.{lin}.bias"] = (c,)
            out[b + "mlp.0.weight"] = (cfg.block_exp * c, c)
            out[b + "mlp.0.bias"] = (cfg.block_exp * c,)
            out[b + "mlp.2.weight"] = (c, cfg.block_exp * c)
            out[b + "mlp.2.bias"] = (c,)
        out[p + "ln_f.weight"] = (c,)
        out[p + "ln_f.bias"] = (c,)

    for idx, (o, i) in zip((0, 2, 4), ((256, 512), (128, 256), (64, 128))):
        out[f"join.{idx}.weight"] = (o, i)
        out[f"join.{idx}.bias"] = (o,)
    if cfg.gru_head:  # nn.GRUCell(64, 64) + nn.Linear(64, 64), model2_seq_30to5.py:842-843
        out["decoder.weight_ih"] = (192, 64)
        out["decoder.weight_hh"] = (192, 64)
        out["decoder.bias_ih"] = (192,)
        out["decoder.bias_hh"] = (192,)
        out["output.weight"] = (64, 64)
        out["output.bias"] = (64,)
    return out


def is_buffer(name: str) -> bool:
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


def _name_seed(name: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:7], "little")


def make_state(cfg: RefConfig, seed: int = 0, scheme: str = "test") -> "OrderedDict[str, torch.Tensor]":
    """Deterministic, name-hashed weights (each tensor has its own CPU generator stream, so the
    table is identical on every host with the same torch build).

    scheme "test": every term is exercised (non-zero biases/pos_emb, BN gamma around 1, non-trivial
    running stats) at magnitudes that keep activations O(1) through ~100 layers.
    scheme "init": the reference's own init (model2_seq.py:207-214 for GPT; torch defaults for
    Linear; ResNet: kaiming-normal fan_out convs, BN (1,0)) - pretrained ImageNet weights
    (model2_seq.py:23,59) need a network fetch and are unavailable offline.
    """
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        g = torch.Generator().manual_seed(_name_seed(name, seed))
        if name.endswith("num_batches_tracked"):
            t = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            t = torch.zeros(shape) if scheme == "init" else 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("running_var"):
            t = torch.ones(shape) if scheme == "init" else 1.0 + 0.2 * torch.rand(shape, generator=g)
        elif len(shape) == 4:  # conv OIHW
            fan_out = shape[0] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        elif ".bn" in name or "downsample.1" in name:
            if scheme == "init":
                t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            elif name.endswith("weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                t = 0.1 * torch.randn(shape, generator=g)
        elif "pos_emb" in name:
            t = torch.zeros(shape) if scheme == "init" else 0.05 * torch.randn(shape, generator=g)
        elif ".ln" in name:
            if scheme == "init":
                t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            elif name.endswith("weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                t = 0.05 * torch.randn(shape, generator=g)
        elif "transformer" in name:  # GPT Linear: N(0, 0.02), bias 0 (model2_seq.py:207-211)
            if name.endswith("weight"):
                t = 0.02 * torch.randn(shape, generator=g)
            else:
                t = torch.zeros(shape) if scheme == "init" else 0.02 * torch.randn(shape, generator=g)
        elif name.startswith("decoder."):  # nn.GRUCell default: U(-1/sqrt(hidden), 1/sqrt(hidden)) for all four tensors
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(64)
        else:  # vel_emb / join / output: torch Linear default U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            fan_in = shape[1] if name.endswith("weight") else None
            if fan_in is None:
                wshape = param_shapes(cfg)[name[: -len("bias")] + "weight"]
                fan_in = wshape[1]
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        sd[name] = t.contiguous()
    return sd


# ----------------------------------------------------------------------------------------------
# synthetic inputs - same generator for CPU oracle and GPU build
# ----------------------------------------------------------------------------------------------
def make_inputs(cfg: RefConfig, batch: int, seed: int = 100, learnable: bool = False):
    """Returns (image_list, lidar_list, radar_list, gps, soft_target, beamidx).

    images: integers 0..255 as fp32 (data2_seq.py:141); lidar BEV: {0,.2,..,1}, mostly zero
    (data2_seq.py:204-211); radar: U[0,1] (Radar_data_preprocessing.py:22-23); gps: angle in
    radians duplicated over the last dim (data2_seq.py:273-280); target: 1.25*N(k;beam,0.5) on
    beam-5..beam+5 (data2_seq.py:162-167).
    """
    g = torch.Generator().manual_seed(seed)
    radar_c = 2 if cfg.add_velocity else 1
    beamidx = torch.randint(0, 64, (batch,), generator=g)
    imgs, lids, rads = [], [], []
    for _ in range(cfg.seq_len * cfg.n_views):
        imgs.append(torch.randint(0, 256, (batch, 3, 256, 256), generator=g).float())
    for _ in range(cfg.seq_len):
        occ = (torch.rand(batch, 1, 256, 256, generator=g) < 0.05).float()
        lvl = torch.randint(1, 6, (batch, 1, 256, 256), generator=g).float() / 5.0
        lids.append(occ * lvl)
        rads.append(torch.rand(batch, radar_c, 256, 256, generator=g))
    if learnable:
        # encode the beam as a bright column band in every modality and as the GPS angle
        for b in range(batch):
            c0 = int(beamidx[b]) * 4
            for t in imgs:
                t[b, :, :, c0:c0 + 4] = 255.0
            for t in lids:
                t[b, :, :, c0:c0 + 4] = 1.0
            for t in rads:
                t[b, :, :, c0:c0 + 4] = 1.0
        ang = (beamidx.float() / 63.0 - 0.5) * math.pi
        ang = ang[:, None].repeat(1, 2) + 0.01 * torch.randn(batch, 2, generator=g)
    else:
        ang = (torch.rand(batch, 2, generator=g) - 0.5) * math.pi
    gps = ang[:, :, None].repeat(1, 1, 2).contiguous()  # (B,2,2), both columns equal
    target = soft_beam_target(beamidx)
    return imgs, lids, rads, gps, target, beamidx


def soft_beam_target(beamidx: torch.Tensor) -> torch.Tensor:
    """data2_seq.py:159-167: 1.25 * normal pdf (sigma .5) on [beam-5, beam+5] ∩ [0,63]."""
    k = torch.arange(64, dtype=torch.float64)[None, :]
    mu = beamidx.to(torch.float64)[:, None]
    pdf = torch.exp(-0.5 * ((k - mu) / 0.5) ** 2) / (0.5 * math.sqrt(2 * math.pi))
    mask = ((k >= (mu - 5).clamp(min=0)) & (k <= (mu + 5).clamp(max=63))).to(torch.float64)
    return (1.25 * pdf * mask).float()


# ----------------------------------------------------------------------------------------------
# functional forward
# ----------------------------------------------------------------------------------------------
class Ctx:
    """Run options + optional capture of intermediate tensors by name."""

    def __init__(self, training: bool = True, dropout: bool = False, update_bn: bool = True,
                 capture: dict | None = None, mask_fn=None):
        self.training = training
        self.dropout = dropout and training
        self.update_bn = update_bn
        self.capture = capture
        # mask_fn(shape, p) -> keep mask already scaled by 1/(1-p), called once per dropout site in forward order
        # (embd_drop of a stage, then per block attn_drop, resid_drop after proj, resid_drop after the MLP); lets a test
        # run the oracle on the very masks the HIP path drew instead of torch's Philox stream
        self.mask_fn = mask_fn

    def tap(self, name, t):
        if self.capture is not None:
            self.capture[name] = t


def normalize_imagenet(x: torch.Tensor) -> torch.Tensor:
    """model2_seq.py:36-45 (input untouched, new tensor returned)."""
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x / 255.0 - mean) / std


def _bn(sd, p, x, ctx: Ctx):
    rm, rv = sd[p + "running_mean"], sd[p + "running_var"]
    if ctx.training and not ctx.update_bn:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, sd[p + "weight"], sd[p + "bias"], training=ctx.training,
                     momentum=0.1, eps=1e-5)
    if ctx.training and ctx.update_bn:
        sd[p + "num_batches_tracked"] += 1
    return y


def _basic_block(sd, p, x, stride, ctx):
    out = F.conv2d(x, sd[p + "conv1.weight"], None, stride, 1)
    out = F.relu(_bn(sd, p + "bn1.", out, ctx))
    out = F.conv2d(out, sd[p + "conv2.weight"], None, 1, 1)
    out = _bn(sd, p + "bn2.", out, ctx)
    if (p + "downsample.0.weight") in sd:
        idn = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride, 0)
        idn = _bn(sd, p + "downsample.1.", idn, ctx)
    else:
        idn = x
    return F.relu(out + idn)


def _stem(sd, p, x, ctx):
    x = F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3)
    x = F.relu(_bn(sd, p + "bn1.", x, ctx))
    return F.max_pool2d(x, 3, 2, 1)


def _layer(sd, p, arch, li, x, ctx):
    for bi in range(RESNET_LAYERS[arch][li - 1]):
        stride = 2 if (li > 1 and bi == 0) else 1
        x = _basic_block(sd, f"{p}layer{li}.{bi}.", x, stride, ctx)
    return x


def _drop(x, p, ctx):
    if not (ctx.dropout and p > 0):
        return x
    if ctx.mask_fn is not None:
        return x * ctx.mask_fn(tuple(x.shape), p)
    return F.dropout(x, p, True)


def _self_attention(sd, p, x, cfg, ctx):
    """model2_seq.py:93-110."""
    B, T, C = x.shape
    nh = cfg.n_head
    hd = C // nh

    def heads(name):
        return F.linear(x, sd[p + name + ".weight"], sd[p + name + ".bias"]).view(B, T, nh, hd).transpose(1, 2)

    k, q, v = heads("key"), heads("query"), heads("value")
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))
    att = _drop(torch.softmax(att, dim=-1), cfg.attn_pdrop, ctx)
    y = (att @ v).transpose(1, 2).reshape(B, T, C)
    y = F.linear(y, sd[p + "proj.weight"], sd[p + "proj.bias"])
    return _drop(y, cfg.resid_pdrop, ctx)


def _block(sd, p, x, cfg, ctx):
    """model2_seq.py:128-134 (pre-LN; MLP activation is ReLU, :123)."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[p + "ln1.weight"], sd[p + "ln1.bias"], 1e-5)
    x = x + _self_attention(sd, p + "attn.", h, cfg, ctx)
    h = F.layer_norm(x, (C,), sd[p + "ln2.weight"], sd[p + "ln2.bias"], 1e-5)
    h = F.relu(F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
    h = _drop(F.linear(h, sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"]), cfg.resid_pdrop, ctx)
    return x + h


def gpt_stage(sd, p, img, lid, rad, gps, cfg, ctx):
    """model2_seq.py:248-287. img/lid/rad: (B*S, C, 8, 8); gps: (B, 2, C).

    Token order inside a sample: image frames t=0..S-1 (64 anchors each, row-major h,w), then
    LiDAR frames, then radar frames, then the two GPS rows.
    """
    S, va, ha = cfg.seq_len, cfg.vert_anchors, cfg.horz_anchors
    C = img.shape[1]
    B = lid.shape[0] // S

    def tok(t, n):
        return t.view(B, n, C, va * ha).permute(0, 1, 3, 2).reshape(B, n * va * ha, C)

    x = torch.cat([tok(img, cfg.n_views * S), tok(lid, S), tok(rad, S), gps], dim=1)
    x = _drop(x + sd[p + "pos_emb"], cfg.embd_pdrop, ctx)
    for i in range(cfg.n_layer):
        x = _block(sd, f"{p}blocks.{i}.", x, cfg, ctx)
    x = F.layer_norm(x, (C,), sd[p + "ln_f.weight"], sd[p + "ln_f.bias"], 1e-5)
    n_sp = (cfg.n_views + 2) * S * va * ha
    gps_out = x[:, n_sp:, :]
    x = x[:, :n_sp, :].view(B, (cfg.n_views + 2) * S, va, ha, C).permute(0, 1, 4, 2, 3)
    ni = cfg.n_views * S
    img_o = x[:, :ni].reshape(B * ni, C, va, ha)
    lid_o = x[:, ni:ni + S].reshape(B * S, C, va, ha)
    rad_o = x[:, ni + S:].reshape(B * S, C, va, ha)
    return img_o, lid_o, rad_o, gps_out


def encoder_forward(sd, image_list, lidar_list, radar_list, gps, cfg: RefConfig, ctx: Ctx):
    """model2_seq.py:473-597 -> (B, 512)."""
    image_list = [normalize_imagenet(x) for x in image_list]
    B = lidar_list[0].shape[0]
    S = cfg.seq_len
    n_views = len(image_list) // S
    assert n_views == cfg.n_views

    def stack(lst):  # frame index = b*len + t  (model2_seq.py:491-493)
        return torch.stack(lst, dim=1).reshape(B * len(lst), *lst[0].shape[1:])

    feats = [stack(image_list), stack(lidar_list), stack(radar_list)]
    trunks = trunk_prefixes()
    feats = [_stem(sd, p, f, ctx) for (p, _), f in zip(trunks, feats)]
    ctx.tap("stem", feats)
    gps_tok = gps
    for s in range(1, 5):
        feats = [_layer(sd, p, arch, s, f, ctx) for (p, arch), f in zip(trunks, feats)]
        ctx.tap(f"layer{s}", feats)
        emb = [F.adaptive_avg_pool2d(f, (cfg.vert_anchors, cfg.horz_anchors)) for f in feats]
        gps_emb = F.linear(gps_tok, sd[f"encoder.vel_emb{s}.weight"], sd[f"encoder.vel_emb{s}.bias"])
        *outs, gps_tok = gpt_stage(sd, f"encoder.transformer{s}.", emb[0], emb[1], emb[2], gps_emb, cfg, ctx)
        ctx.tap(f"gpt{s}", list(outs) + [gps_tok])
        scale = 8 // (2 ** (s - 1))
        if scale > 1:  # scales 8,4,2; the last stage adds the 8x8 map directly (:577-579)
            outs = [F.interpolate(o, scale_factor=scale, mode="bilinear") for o in outs]
        feats = [f + o for f, o in zip(feats, outs)]
    pooled = [f.mean(dim=(2, 3)).view(B, -1, 512) for f in feats]
    fused = torch.cat(pooled + [gps_tok], dim=1).sum(dim=1)
    ctx.tap("fused", fused)
    return fused


def join_forward(sd, fused):
    """model2_seq.py:863-869."""
    h = F.relu(F.linear(fused, sd["join.0.weight"], sd["join.0.bias"]))
    h = F.relu(F.linear(h, sd["join.2.weight"], sd["join.2.bias"]))
    return F.linear(h, sd["join.4.weight"], sd["join.4.bias"])


def gru_head_forward(sd, z, pred_len):
    """The autoregressive beam-sequence head of model2_seq_30to5.py:846-862: x = 0, hidden = z (the join output);
    pred_len times: hidden = GRUCell(x, hidden); x = x + output(hidden); collect x  ->  (B, pred_len, 64).
    GRUCell is written out (torch.nn.GRUCell documentation: r, z, n gate order); tests pin it against nn.GRUCell."""
    w_ih, w_hh, b_ih, b_hh = (sd["decoder." + k] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
    x = torch.zeros((z.shape[0], 64), dtype=z.dtype)
    h = z
    out = []
    for _ in range(pred_len):
        gi = F.linear(x, w_ih, b_ih)
        gh = F.linear(h, w_hh, b_hh)
        i_r, i_z, i_n = gi.chunk(3, 1)
        h_r, h_z, h_n = gh.chunk(3, 1)
        r = torch.sigmoid(i_r + h_r)
        u = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1 - u) * n + u * h
        x = F.linear(h, sd["output.weight"], sd["output.bias"]) + x
        out.append(x)
    return torch.stack(out, dim=1)


def transfuser_forward(sd, image_list, lidar_list, radar_list, gps, cfg: RefConfig, ctx: Ctx | None = None):
    """TransFuser.forward (model2_seq.py:880-894) with the GPT ``Encoder`` wired in (:860); with cfg.gru_head the
    forward of model2_seq_30to5.py:846-862 (same encoder + join, then the GRU head)."""
    ctx = ctx or Ctx()
    z = join_forward(sd, encoder_forward(sd, image_list, lidar_list, radar_list, gps, cfg, ctx))
    return gru_head_forward(sd, z, cfg.pred_len) if cfg.gru_head else z
