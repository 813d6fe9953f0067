"""Generates tests/golden/fusion30to5_golden.npz by running the REFERENCE's own 30->5 model code (build container only;
/root/reference never travels).

/root/reference/model2_seq_30to5.py imports `torchvision` / `mamba_ssm` (absent here: ordinary ModuleNotFoundError); the
same two stub modules as make_golden.py are pre-inserted (a ResNet-v1 skeleton with torchvision's attribute names, and a
`Mamba` that raises when constructed).  `TransFuser.__init__` (:817-843) wires `EncoderWithMamba`, which needs the CUDA-only
`mamba_ssm`, so the model is assembled from the reference's own pieces instead: `model2_seq_30to5.Encoder(config)` (the GPT
encoder of :405-597, constructed with `config_seq_30to5.GlobalConfig(seq_len=10, ...)` => pos_emb of 1922 tokens, :188),
and `join` / `decoder = nn.GRUCell(64, 64)` / `output = nn.Linear(64, 64)` built exactly as :834-843.  The forward that is
executed is the reference's `TransFuser.forward` itself (:845-862): its FunctionDef is picked out of the class with `ast`
and run with `self` bound to a holder of those modules - every line of the autoregressive head is the reference's.

Weights come from oracle/fusion_ref.py::make_state (strict=True load: proves names and shapes of the seq_len-10 variant),
inputs from make_inputs; only seeds + the reference's outputs are stored.  The oracle is checked against the reference on
the spot (fusion30to5 lines of oracle_vs_reference.txt).
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_30to5.py
"""
import ast
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch
from torch import nn

import make_golden as mg   # the stub installer (nothing else of it runs)
from oracle import fusion_ref as fr
from oracle import train_ref as tr

REF = "/root/reference"
CFG = dict(seq_len=10, n_layer=2, pred_len=5, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
STATE_SEED, INPUT_SEED, TARGET_SEED, BATCH = 21, 300, 7, 2
GRAD_PROBES = ("decoder.weight_hh", "decoder.bias_ih", "output.weight", "join.0.weight", "encoder.transformer1.pos_emb",
               "encoder.transformer4.blocks.1.attn.query.weight", "encoder.vel_emb2.weight",
               "encoder.lidar_encoder._model.layer2.0.downsample.0.weight", "encoder.image_encoder.features.conv1.weight")


def reference_forward():
    """-> the reference's TransFuser.forward (model2_seq_30to5.py:845-862) as a plain function of (self, ...)"""
    path = os.path.join(REF, "model2_seq_30to5.py")
    tree = ast.parse(open(path).read(), filename=path)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TransFuser"]
    assert len(cls) == 1
    fwd = [n for n in cls[0].body if isinstance(n, ast.FunctionDef) and n.name == "forward"]
    assert len(fwd) == 1
    ns = dict(torch=torch)
    exec(compile(ast.Module(body=fwd, type_ignores=[]), path, "exec"), ns)
    return ns["forward"]


def _maxabs(a, b):
    return float((a.detach() - b.detach()).abs().max())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    mg._install_stubs()
    sys.path.insert(0, REF)
    import config_seq_30to5
    import model2_seq_30to5 as m30

    cfg = fr.RefConfig(gru_head=True, **CFG)
    rc = config_seq_30to5.GlobalConfig(add_velocity=cfg.add_velocity, **CFG)
    assert rc.seq_len == 10 and rc.pred_len == 5
    sd = fr.make_state(cfg, seed=STATE_SEED)
    enc = m30.Encoder(rc)
    holder = types.SimpleNamespace(
        encoder=enc,
        join=nn.Sequential(nn.Linear(512, 256), nn.ReLU(inplace=True), nn.Linear(256, 128), nn.ReLU(inplace=True),
                           nn.Linear(128, 64)),                       # :834-840
        decoder=nn.GRUCell(input_size=64, hidden_size=64),            # :842
        output=nn.Linear(64, 64),                                     # :843
        pred_len=rc.pred_len, device=torch.device("cpu"), config=rc)
    assert tuple(enc.transformer1.pos_emb.shape) == (1, 1922, 64)
    for prefix, mod in (("encoder.", enc), ("join.", holder.join), ("decoder.", holder.decoder), ("output.", holder.output)):
        mod.load_state_dict({k[len(prefix):]: v.clone() for k, v in sd.items() if k.startswith(prefix)}, strict=True)
        mod.train()
    forward = reference_forward()

    imgs, lids, rads, gps, _, _ = fr.make_inputs(cfg, BATCH, seed=INPUT_SEED)
    assert len(imgs) == 10
    target = torch.rand(BATCH, 5, 64, generator=torch.Generator().manual_seed(TARGET_SEED)) * 0.5
    pred_ref = forward(holder, imgs, lids, rads, gps)
    assert tuple(pred_ref.shape) == (BATCH, 5, 64)
    loss_ref = tr.sigmoid_focal_loss(pred_ref, target)
    loss_ref.backward()

    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    cap = {}
    pred_my = fr.transfuser_forward(sdo, imgs, lids, rads, gps, cfg, fr.Ctx(training=True, capture=cap))
    loss_my = tr.sigmoid_focal_loss(pred_my, target)
    loss_my.backward()
    ref_named = {}
    for prefix, mod in (("encoder.", enc), ("join.", holder.join), ("decoder.", holder.decoder), ("output.", holder.output)):
        ref_named.update((prefix + k, v) for k, v in mod.named_parameters())
    assert set(ref_named) == {k for k, v in sdo.items() if v.requires_grad}
    report = [("fusion30to5 pred (B=2, seq_len 10, 1922 tokens)", _maxabs(pred_ref, pred_my)),
              ("fusion30to5 loss", abs(float(loss_ref) - float(loss_my))),
              # attn.key.bias gradients are analytically zero (softmax is invariant to a per-query shift of the scores):
              # both sides hold rounding noise of 1e-12 there, so they are compared absolutely, not relative to their maximum
              ("fusion30to5 grads (max rel-to-max over all params)",
               max(_maxabs(v.grad, sdo[k].grad) / (float(v.grad.abs().max()) + 1e-12) for k, v in ref_named.items()
                   if not k.endswith("attn.key.bias"))),
              ("fusion30to5 grads of attn.key.bias (analytically 0; max abs)",
               max(max(float(v.grad.abs().max()), float(sdo[k].grad.abs().max())) for k, v in ref_named.items()
                   if k.endswith("attn.key.bias")))]
    with torch.no_grad():
        fused_ref = enc(imgs, lids, rads, gps)  # second train-mode pass: BN batch statistics => same output, buffers move on
    report.append(("fusion30to5 encoder fused", _maxabs(fused_ref, cap["fused"])))
    for m in (enc, holder.join, holder.decoder, holder.output):
        m.eval()
    with torch.no_grad():
        pred_eval_ref = forward(holder, imgs, lids, rads, gps)

    out = dict(meta=np.array([STATE_SEED, INPUT_SEED, TARGET_SEED, BATCH, CFG["seq_len"], CFG["n_layer"], CFG["pred_len"]]),
               pred_b2=pred_ref.detach().numpy(), fused_b2=fused_ref.numpy(), loss_b2=np.array(float(loss_ref)),
               pred_eval_after_two_passes=pred_eval_ref.numpy())
    for name in GRAD_PROBES:
        gt = ref_named[name].grad
        out["grad:" + name + ":absmax"] = np.array(float(gt.abs().max()))
        out["grad:" + name + ":l2"] = np.array(float(gt.norm()))
        out["grad:" + name + ":head"] = gt.flatten()[:16].numpy().copy()
        out["grad:" + name + ":strided"] = gt.flatten()[::max(1, gt.numel() // 64)][:64].numpy().copy()
    np.savez(os.path.join(HERE, "fusion30to5_golden.npz"), **out)

    log = os.path.join(HERE, "oracle_vs_reference.txt")
    kept = [l for l in open(log).read().splitlines() if not l.startswith("fusion30to5")] if os.path.exists(log) else []
    bad = False
    print("oracle vs reference model2_seq_30to5.py (max abs diff):")
    for k, v in report:
        print(f"  {k:60s} {v:.3e}")
        kept.append(f"{k:60s} {v:.3e}")
        bad = bad or not (v < 5e-5)
    open(log, "w").write("\n".join(kept) + "\n")
    if bad:
        raise SystemExit("oracle does not match reference")


if __name__ == "__main__":
    main()
