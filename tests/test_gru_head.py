"""The 30->5 variant's autoregressive GRU head (model2_seq_30to5.py:842-862): oracle restatement pinned
against torch.nn.GRUCell on CPU; HIP kernels (csrc/gru.hip) and TransFuser30to5 against the oracle on the GPU."""
import pytest
import torch

from oracle import fusion_ref as fr


def _head_state(seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * 0.125  # noqa: E731  (nn.GRUCell init range 1/sqrt(64))
    return {"decoder.weight_ih": r(192, 64), "decoder.weight_hh": r(192, 64), "decoder.bias_ih": r(192),
            "decoder.bias_hh": r(192), "output.weight": r(64, 64), "output.bias": r(64)}


def test_oracle_gru_head_matches_torch_modules():
    """the written-out recurrence == the reference's module code run with torch's own nn.GRUCell / nn.Linear"""
    sd = _head_state(1)
    dec = torch.nn.GRUCell(64, 64)
    out = torch.nn.Linear(64, 64)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")})
    out.load_state_dict({"weight": sd["output.weight"], "bias": sd["output.bias"]})
    z = torch.randn(5, 64, generator=torch.Generator().manual_seed(2))
    x = torch.zeros(5, 64)
    h, wp = z, []
    for _ in range(5):  # model2_seq_30to5.py:853-860
        h = dec(x, h)
        x = out(h) + x
        wp.append(x)
    want = torch.stack(wp, dim=1)
    got = fr.gru_head_forward(sd, z, 5)
    assert got.shape == (5, 5, 64)
    assert torch.allclose(got, want.detach(), rtol=1e-5, atol=2e-6), (got - want.detach()).abs().max()


@pytest.mark.gpu
def test_gru_head_kernels_vs_oracle(dev):
    from deepsense6g_tii_amd._lib import lib
    L = lib()
    B, T = 12, 5
    sd = _head_state(3)
    z = torch.randn(B, 64, generator=torch.Generator().manual_seed(4))
    dpred = torch.randn(B, T, 64, generator=torch.Generator().manual_seed(5))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    zr = z.clone().requires_grad_(True)
    want = fr.gru_head_forward(sdr, zr, T)
    want.backward(dpred)
    d = {k: v.to(dev) for k, v in sd.items()}
    zg, dpg = z.to(dev), dpred.to(dev)
    pred = torch.empty((B, T, 64), device=dev)
    saved = torch.empty(L.gru_head_saved_floats(B, T), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    args = [d["decoder.weight_ih"], d["decoder.weight_hh"], d["decoder.bias_ih"], d["decoder.bias_hh"], d["output.weight"],
            d["output.bias"]]
    L.gru_head_fwd(zg.data_ptr(), *[a.data_ptr() for a in args], pred.data_ptr(), saved.data_ptr(), B, T, 64, st)
    assert torch.allclose(pred.cpu(), want.detach(), rtol=1e-5, atol=1e-6)
    npar = L.gru_head_slab_floats()
    assert npar == sum(v.numel() for v in sd.values())
    slabs = torch.empty((B, npar), device=dev)
    dz = torch.empty((B, 64), device=dev)
    L.gru_head_bwd(dpg.data_ptr(), zg.data_ptr(), saved.data_ptr(), args[0].data_ptr(), args[1].data_ptr(),
                   args[4].data_ptr(), dz.data_ptr(), slabs.data_ptr(), B, T, 64, st)
    assert torch.allclose(dz.cpu(), zr.grad, rtol=1e-4, atol=1e-6)
    tot = slabs.sum(0).cpu()
    off = 0
    for k in ("decoder.weight_ih", "decoder.weight_hh", "decoder.bias_ih", "decoder.bias_hh", "output.weight", "output.bias"):
        n = sd[k].numel()
        got = tot[off:off + n].view(sd[k].shape)
        off += n
        assert torch.allclose(got, sdr[k].grad, rtol=1e-4, atol=1e-5), k


@pytest.mark.gpu
def test_transfuser_30to5_vs_oracle(dev):
    """TransFuser30to5 (GRU head on the fusion path) against the oracle: predictions (B, pred_len, 64), loss and the
    gradients of head / join / a GPT parameter; state-dict names of the reference variant."""
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser30to5
    from oracle import train_ref as tr
    kw = dict(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, n_layer=1, seq_len=2, pred_len=5)
    rcfg = fr.RefConfig(gru_head=True, **kw)
    sd = fr.make_state(rcfg, seed=9)
    model = TransFuser30to5(GlobalConfig(**kw), dev)
    assert {"decoder.weight_ih", "decoder.weight_hh", "decoder.bias_ih", "decoder.bias_hh", "output.weight",
            "output.bias"} <= set(model.state_dict().keys())
    model.load_state_dict(sd)
    model.train()
    imgs, lids, rads, gps, _, _ = fr.make_inputs(rcfg, 2, seed=100)
    target = torch.rand(2, 5, 64, generator=torch.Generator().manual_seed(1)) * 0.5
    loss, pred = model.train_step_loss(imgs, lids, rads, gps, target)
    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    want = fr.transfuser_forward(sdo, imgs, lids, rads, gps, rcfg, fr.Ctx(training=True))
    lref = tr.sigmoid_focal_loss(want, target)
    lref.backward()
    assert pred.shape == (2, 5, 64)
    assert (pred.cpu() - want.detach()).abs().max().item() <= 1e-3 * want.detach().abs().max().item()
    assert abs(float(loss) - float(lref)) <= 1e-5 * max(1.0, abs(float(lref)))
    params = dict(model.named_parameters())
    for name in ("decoder.weight_hh", "decoder.bias_ih", "output.weight", "join.4.weight", "join.0.bias",
                 "encoder.transformer4.blocks.0.mlp.0.weight"):
        g, gr = params[name].grad.cpu(), sdo[name].grad
        rel = (g - gr).norm().item() / (gr.norm().item() + 1e-30)
        assert rel < 2e-2, (name, rel)   # fp32 gradients through ReLU / max-pool decisions: see test_model_gpu
    # eval-mode inference path (no saved state)
    model.eval()
    with torch.no_grad():
        out = model(imgs, lids, rads, gps)
    assert out.shape == (2, 5, 64) and torch.isfinite(out).all()
