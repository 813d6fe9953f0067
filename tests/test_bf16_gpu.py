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


def test_model_bf16_gradients_against_the_bf16_rounding_floor(dev, bf16_mode):
    """g1 is "parity unpinned" (the reference has no reduced-precision path), so the bar is a measured one: an
    INDEPENDENT bf16 forward / backward of the same network - the CPU oracle under torch.autocast(bfloat16): conv / linear /
    matmul on bf16 operands with bf16 outputs, norms and softmax in fp32 - gives the rounding floor of bf16 arithmetic on
    this problem, per tensor, against an fp64 run of the oracle.  The HIP bf16-storage path must sit on that floor for
    EVERY parameter tensor (~170 here), not just point the same way on three of them: median L2-relative error within
    1.5x of the autocast run's, the share of tensors whose gradient keeps cosine > 0.9 with fp64 no smaller, worst
    tensor no worse than 2x the autocast run's worst, logits within 2x of its logit error."""
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    from tests.test_model_gpu import _oracle_fp64_grads
    kw = dict(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, n_layer=2)
    rcfg = fr.RefConfig(**kw)
    sd = fr.make_state(rcfg, seed=3)
    model = TransFuser(GlobalConfig(**kw), dev)
    model.load_state_dict(sd)
    model.train()
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 2, seed=100)
    loss, logits = model.train_step_loss(imgs, lids, rads, gps, target)
    torch.cuda.synchronize()
    assert model._use16, "the bf16-storage path must be the one that ran"
    hip = {n: p.grad.detach().cpu() for n, p in model.named_parameters()}
    # fp32 oracle (logit reference), bf16-autocast oracle (the floor), fp64 oracle (the yardstick)
    with torch.no_grad():
        ref = fr.transfuser_forward({k: v.clone() for k, v in sd.items()}, imgs, lids, rads, gps, rcfg, fr.Ctx(training=True))
    sda = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lg16 = fr.transfuser_forward(sda, imgs, lids, rads, gps, rcfg, fr.Ctx(training=True))
    tr.sigmoid_focal_loss(lg16.float(), target).backward()
    g64 = _oracle_fp64_grads(sd, rcfg, imgs, lids, rads, gps, target)

    def stats(grads):
        l2, cos = [], []
        for k, r64 in g64.items():
            n = r64.norm().item()
            if r64.abs().max().item() < 1e-12:      # attn.key.bias: mathematically zero
                continue
            g = grads[k].double()
            l2.append(((g - r64).norm().item() / n, k))
            cos.append((torch.dot(g.flatten(), r64.flatten()) / (g.norm() * r64.norm() + 1e-300)).item())
        l2.sort(reverse=True)
        return l2, sum(c > 0.9 for c in cos) / len(cos)

    h_l2, h_cos = stats(hip)
    a_l2, a_cos = stats({k: v.grad for k, v in sda.items() if isinstance(v, torch.Tensor) and v.requires_grad})
    med = lambda v: v[len(v) // 2][0]  # noqa: E731
    e_hip, e_ac = relerr(logits, ref), relerr(lg16.detach().float(), ref)
    print(f"bf16 floor over {len(h_l2)} tensors (L2-relative vs fp64): HIP median {med(h_l2):.3f} worst {h_l2[0][0]:.3f} "
          f"({h_l2[0][1]}) cos>0.9 {h_cos:.2f} | autocast oracle median {med(a_l2):.3f} worst {a_l2[0][0]:.3f} cos>0.9 {a_cos:.2f}"
          f" | logits vs fp32: HIP {e_hip:.2e} autocast {e_ac:.2e}", flush=True)
    assert len(h_l2) > 150
    assert med(h_l2) < 1.5 * med(a_l2) + 0.02
    assert h_cos > a_cos - 0.05
    assert h_l2[0][0] < 2.0 * a_l2[0][0]
    assert e_hip < 2.0 * e_ac + 2e-3
    assert all(torch.isfinite(g).all() for g in hip.values())


# ---- split-bf16 mode (ds6g_set_compute_mode(2), "f32x3"): a*b = hi*hi + hi*lo + lo*hi on the bf16 matrix cores -------
# Held to the bar of the exact path (north_star: beam logits within 1e-3 relative of the fp32 reference), per kernel
# to ~2^-16 of the largest output element.

@pytest.fixture()
def x3_mode():
    from deepsense6g_tii_amd import ops
    ops.set_compute_mode("f32x3")
    assert ops.get_compute_mode() == "f32x3"
    yield
    ops.set_compute_mode("f32")
    assert ops.get_compute_mode() == "f32"


def test_compute_mode_rejects_unknown(dev):
    from deepsense6g_tii_amd import ops
    from deepsense6g_tii_amd._lib import Ds6gError, lib
    with pytest.raises(ValueError):
        ops.set_compute_mode("fp8")
    with pytest.raises(Ds6gError):
        lib().set_compute_mode(4)
    assert ops.get_compute_mode() == "f32"


def test_conv_and_linear_x3(dev, x3_mode):
    from deepsense6g_tii_amd import ops
    g = torch.Generator().manual_seed(1)
    ws = ops.Workspace(dev, 64 << 20)
    N, H, C, K = 4, 16, 64, 128
    x = torch.randn(N, C, H, H, generator=g, requires_grad=True)
    w = (torch.randn(K, C, 3, 3, generator=g) / math.sqrt(C * 9)).requires_grad_(True)
    dy = torch.randn(N, K, H, H, generator=g)
    F.conv2d(x, w, None, 1, 1).backward(dy)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    wg = w.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    y = ops.conv2d_fwd(xg, wg.data_ptr(), K, 3, 3, 1, 1).cpu().permute(0, 3, 1, 2)
    errs = {"fwd": relerr(y, F.conv2d(x, w, None, 1, 1).detach())}
    dx = ops.conv2d_dgrad(dyg, wg.data_ptr(), tuple(xg.shape), 3, 3, 1, 1).cpu().permute(0, 3, 1, 2)
    errs["dgrad"] = relerr(dx, x.grad)
    dw = torch.empty_like(wg)
    ops.conv2d_wgrad(xg, dyg, dw.data_ptr(), 3, 3, 1, 1, ws)
    errs["wgrad"] = relerr(dw.cpu().permute(0, 3, 1, 2), w.grad)
    M, Nn, Kk = 962, 256, 512
    a = torch.randn(M, Kk, generator=g)
    wl = torch.randn(Nn, Kk, generator=g) / math.sqrt(Kk)
    b = torch.randn(Nn, generator=g)
    wlg, bg = wl.to(dev), b.to(dev)
    yl = ops.linear_fwd(a.to(dev), wlg.data_ptr(), bg.data_ptr(), Nn, relu=True)
    errs["linear"] = relerr(yl, F.relu(F.linear(a, wl, b)))
    print("x3 kernel errors", errs)
    assert max(errs.values()) < 2e-5, errs   # ~2^-16; the bf16 mode sits at ~5e-3 on the same data


@pytest.mark.parametrize("hd", [16, 32, 64, 128])
def test_attention_x3(dev, x3_mode, hd):
    from deepsense6g_tii_amd import ops
    B, T, nh = 2, 962, 4
    C = nh * hd
    g = torch.Generator().manual_seed(hd + 1)
    q, k, v, do = (torch.randn(B * T, C, generator=g) for _ in range(4))
    ws = ops.Workspace(dev, 512 << 20)

    def heads(t):
        return t.view(B, T, nh, hd).transpose(1, 2)

    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    att = torch.softmax((heads(qr) @ heads(kr).transpose(-2, -1)) / math.sqrt(hd), dim=-1)
    o_ref = (att @ heads(vr)).transpose(1, 2).reshape(B * T, C)
    o_ref.backward(do)
    qg, kg, vg, dog = q.to(dev), k.to(dev), v.to(dev), do.to(dev)
    o, lse = ops.attention_fwd(qg, kg, vg, B, T, nh, ws)
    dq, dk, dv = ops.attention_bwd(qg, kg, vg, o, dog, lse, B, T, nh, ws)
    errs = dict(o=relerr(o, o_ref.detach()), dq=relerr(dq, qr.grad), dk=relerr(dk, kr.grad), dv=relerr(dv, vr.grad))
    print("x3 attention errors", hd, errs)
    assert max(errs.values()) < 1e-4, errs


@pytest.mark.parametrize("mode", ["f32x3", "f32x6"])
def test_model_split_modes_meet_fp32_bar(dev, mode):
    """Whole path in the split-bf16 modes against the fp32 oracle on identical weights / inputs: logits and loss within
    1e-3 relative (the north_star bar of the exact path; measured 1.3e-5 / 2e-7 for f32x3).  Parameter gradients are
    ill-conditioned in fp32 at this batch of 2 (ReLU / max-pool decisions, BN cancellation: the fp32 CPU oracle itself is
    several percent off an fp64 run on some tensors), so they are held to the yardstick of tests/test_model_gpu.py,
    deviation from the fp64 oracle relative to the fp32 oracle's own: f32x6 (24 significand bits) within 3x like the exact
    path; f32x3 (16 bits: per-kernel error 4.7e-6 vs 1.1e-6) within 15x on the median (measured 8x: 1.5e-2 vs 1.7e-3)."""
    from deepsense6g_tii_amd import ops
    ops.set_compute_mode(mode)
    try:
        _split_mode_model_check(dev, 3.0 if mode == "f32x6" else 15.0)
    finally:
        ops.set_compute_mode("f32")


def _split_mode_model_check(dev, grad_factor):
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    from tests.test_model_gpu import _oracle_fp64_grads
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
    lref = tr.sigmoid_focal_loss(ref, target)
    lref.backward()
    e_logit = relerr(logits, ref.detach())
    e_loss = abs(float(loss) - float(lref)) / abs(float(lref))
    g64 = _oracle_fp64_grads(sd, rcfg, imgs, lids, rads, gps, target)
    grads = {n: p.grad.cpu() for n, p in model.named_parameters()}
    hip, o32 = [], []
    for k, r64 in g64.items():
        scale = r64.abs().max().item()
        if scale < 1e-12:  # attn.key.bias: mathematically zero
            assert grads[k].abs().max().item() < 1e-6, k
            continue
        hip.append(((grads[k].double() - r64).abs().max().item() / scale, k))
        o32.append(((sdo[k].grad.double() - r64).abs().max().item() / scale, k))
    hip.sort(reverse=True)
    o32.sort(reverse=True)
    med = lambda v: v[len(v) // 2][0]
    print(f"split model: logits {e_logit:.2e}  loss {e_loss:.2e}  grad err vs fp64: x3 median {med(hip):.2e} max "
          f"{hip[0][0]:.2e} ({hip[0][1]}) | oracle32 median {med(o32):.2e} max {o32[0][0]:.2e}")
    assert e_logit < 1e-3 and e_loss < 1e-3
    assert med(hip) < grad_factor * med(o32) + 1e-4
    assert hip[0][0] < max(grad_factor * o32[0][0], 0.05)
