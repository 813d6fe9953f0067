"""GPU: the bf16 matrix-core mode (ds6g_set_compute_mode(1): operands rounded to bf16 on the way into the MFMA,
fp32 accumulate, fp32 storage).  The reference has no mixed precision, so this throughput
configuration is validated against fp32 torch with builder-declared tolerances: a kernel must agree with the same
product computed in fp32 on bf16-ROUNDED operands to fp32 accuracy (it IS that product), and with the unrounded
fp32 product to bf16 accuracy."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16_mode():
    from deepsense6g_tii_amd._lib import lib
    lib().set_compute_mode(1)
    assert lib().get_compute_mode() == 1
    yield
    lib().set_compute_mode(0)
    assert lib().get_compute_mode() == 0


def rb(t):  # round to bf16 and back (RNE), what the kernels do to MFMA operands
    return t.to(torch.bfloat16).float()


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()


def test_conv_and_linear_bf16(dev, bf16_mode):
    from deepsense6g_tii_amd import ops
    g = torch.Generator().manual_seed(0)
    ws = ops.Workspace(dev, 64 << 20)
    N, H, C, K = 4, 16, 64, 128
    x = torch.randn(N, C, H, H, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / math.sqrt(C * 9)
    dy = torch.randn(N, K, H, H, generator=g)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wg = w.permute(0, 2, 3, 1).contiguous().to(dev)
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    # forward
    y = ops.conv2d_fwd(xg, wg.data_ptr(), K, 3, 3, 1, 1).cpu().permute(0, 3, 1, 2)
    assert relerr(y, F.conv2d(rb(x), rb(w), None, 1, 1)) < 2e-5          # exactly the bf16-operand product
    assert relerr(y, F.conv2d(x, w, None, 1, 1)) < 2e-2                    # bf16 accuracy vs the fp32 product
    # dgrad / wgrad
    xr, wr = rb(x).requires_grad_(True), rb(w).requires_grad_(True)
    F.conv2d(xr, rb(w), None, 1, 1).backward(rb(dy))
    dx = ops.conv2d_dgrad(dyg, wg.data_ptr(), tuple(xg.shape), 3, 3, 1, 1).cpu().permute(0, 3, 1, 2)
    assert relerr(dx, xr.grad) < 2e-5
    F.conv2d(rb(x), wr, None, 1, 1).backward(rb(dy))
    dw = torch.empty_like(wg)
    ops.conv2d_wgrad(xg, dyg, dw.data_ptr(), 3, 3, 1, 1, ws)
    assert relerr(dw.cpu().permute(0, 3, 1, 2), wr.grad) < 5e-5
    # linear with bias/relu epilogue (bias is added in fp32)
    M, Nn, Kk = 962, 256, 64
    a = torch.randn(M, Kk, generator=g)
    wl = torch.randn(Nn, Kk, generator=g) / 8
    b = torch.randn(Nn, generator=g)
    wlg, bg = wl.to(dev), b.to(dev)   # keep the device copies alive across the call (raw pointers)
    yl = ops.linear_fwd(a.to(dev), wlg.data_ptr(), bg.data_ptr(), Nn, relu=True)
    assert relerr(yl, F.relu(F.linear(rb(a), rb(wl), b))) < 2e-5


@pytest.mark.parametrize("hd", [16, 32, 64, 128])
def test_attention_bf16(dev, bf16_mode, hd):
    from deepsense6g_tii_amd import ops
    B, T, nh = 2, 962, 4
    C = nh * hd
    g = torch.Generator().manual_seed(hd)
    q, k, v, do = (torch.randn(B * T, C, generator=g) for _ in range(4))
    ws = ops.Workspace(dev, 512 << 20)

    def ref(q, k, v):
        def heads(t):
            return t.view(B, T, nh, hd).transpose(1, 2)
        att = torch.softmax((heads(q) @ heads(k).transpose(-2, -1)) / math.sqrt(hd), dim=-1)
        return (att @ heads(v)).transpose(1, 2).reshape(B * T, C)

    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref = ref(qr, kr, vr)
    o_ref.backward(do)
    qg, kg, vg, dog = q.to(dev), k.to(dev), v.to(dev), do.to(dev)
    o, lse = ops.attention_fwd(qg, kg, vg, B, T, nh, ws)
    assert relerr(o, o_ref.detach()) < 2e-2
    dq, dk, dv = ops.attention_bwd(qg, kg, vg, o, dog, lse, B, T, nh, ws)
    assert relerr(dq, qr.grad) < 3e-2
    assert relerr(dk, kr.grad) < 3e-2
    assert relerr(dv, vr.grad) < 3e-2


def test_model_bf16_close_to_fp32_oracle_and_learns(dev, bf16_mode):
    """Whole path in bf16 matrix-core mode: logits within 3e-2 (relative to the largest logit) of the fp32 oracle
    on identical weights/inputs, finite gradients that point the same way as the fp32 ones, and the fused
    training step reduces the loss."""
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from deepsense6g_tii_amd.train import FusedAdamW, train_iteration
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    kw = dict(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, n_layer=2)
    rcfg = fr.RefConfig(**kw)
    sd = fr.make_state(rcfg, seed=3)
    model = TransFuser(GlobalConfig(**kw), dev)
    model.load_state_dict(sd)
    model.train()
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 2, seed=100)
    loss, logits = model.train_step_loss(imgs, lids, rads, gps, target)
    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    ref = fr.transfuser_forward(sdo, imgs, lids, rads, gps, rcfg, fr.Ctx(training=True))
    tr.sigmoid_focal_loss(ref, target).backward()
    assert relerr(logits, ref.detach()) < 3e-2
    for name in ("join.4.weight", "encoder.transformer4.blocks.1.mlp.0.weight", "encoder.vel_emb1.weight"):
        g = dict(model.named_parameters())[name].grad.cpu().flatten()
        gr = sdo[name].grad.flatten()
        assert torch.isfinite(g).all()
        assert (torch.dot(g, gr) / (g.norm() * gr.norm())).item() > 0.9, name   # cosine with the fp32 gradient
    opt = FusedAdamW(model, lr=2e-4)
    losses = []
    for _ in range(6):
        l, _ = train_iteration(model, opt, (imgs, lids, rads, gps, target))
        losses.append(float(l))
    assert losses[-1] < losses[0]
