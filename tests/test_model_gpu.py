"""End-to-end parity of the HIP path (through the TransFuser boundary and the C ABI) against
(a) the golden fixture generated from the REFERENCE code (tests/golden/fusion_golden.npz) and
(b) the CPU oracle run on the same seeded inputs, stage by stage.
Tolerance: 1e-3 relative (max|a-b| / max|b|), the north-star bound for fp32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fusion_golden.npz")
TOL = 1e-3


def _host_threads():
    # the box's cgroup share, not the host's core count (oversubscribed OpenMP crawls)
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _build(dev, cfg_kw, seed):
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from oracle import fusion_ref as fr
    rcfg = fr.RefConfig(**cfg_kw)
    sd = fr.make_state(rcfg, seed=seed)
    model = TransFuser(GlobalConfig(**cfg_kw), dev)
    missing = model.load_state_dict(sd, strict=True)
    assert model.params_in_arena()
    return model, rcfg, sd


def _nchw(t):
    return t.cpu().permute(0, 3, 1, 2)


@pytest.fixture(scope="module")
def run_b2(dev):
    """One forward+backward of the HIP path and of the oracle on identical inputs/weights (B=2)."""
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    import time
    t0 = time.time()

    def tick(msg):
        print(f"[run_b2 {time.time() - t0:7.1f}s] {msg}", flush=True)

    kw = dict(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    model, rcfg, sd = _build(dev, kw, seed=3)
    tick("model built")
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 2, seed=100)
    tick("inputs made")
    model.train()
    cap = {}
    model._capture = cap
    loss, logits = model.train_step_loss(imgs, lids, rads, gps, target)
    model._capture = None
    torch.cuda.synchronize()
    tick("HIP forward+backward done")
    grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
    bufs = {n: b.detach().cpu().clone() for n, b in model.named_buffers()}
    # eval-mode forward on the updated running stats
    model.eval()
    with torch.no_grad():
        logits_eval = model(imgs, lids, rads, gps).cpu()
    # oracle
    torch.set_num_threads(_host_threads())
    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    ocap = {}
    ologits = fr.transfuser_forward(sdo, imgs, lids, rads, gps, rcfg, fr.Ctx(training=True, capture=ocap))
    oloss = tr.sigmoid_focal_loss(ologits, target)
    oloss.backward()
    tick("oracle forward+backward done")
    return dict(model=model, loss=loss.cpu(), logits=logits.cpu(), grads=grads, bufs=bufs, cap=cap, ocap=ocap,
                ologits=ologits.detach(), oloss=oloss.detach(), sdo=sdo, logits_eval=logits_eval)


def test_forward_matches_reference_golden(run_b2):
    gold = np.load(GOLD)
    r = run_b2
    report = []
    for name in ("stem", "layer1", "layer2", "layer3", "layer4"):
        for m in range(3):
            report.append((f"{name}[{m}]", rel(_nchw(r["cap"][name][m]), r["ocap"][name][m].detach())))
    report.append(("fused", rel(r["cap"]["fused"], r["ocap"]["fused"].detach())))
    msg = "\n".join(f"{k:12s} {v:.3e}" for k, v in report)
    print(msg)
    assert rel(r["logits"], gold["logits_b2"]) < TOL, msg
    assert rel(r["cap"]["fused"], gold["fused_b2"]) < TOL, msg
    assert rel(r["logits"], r["ologits"]) < TOL
    assert abs(float(r["loss"]) - float(gold["loss_b2"])) < TOL * float(gold["loss_b2"])
    assert max(v for _, v in report) < TOL, msg


def test_eval_mode_matches_reference_golden(run_b2):
    gold = np.load(GOLD)
    assert rel(run_b2["logits_eval"], gold["logits_eval_b2"]) < TOL


def test_bn_running_stats_match(run_b2):
    gold = np.load(GOLD)
    b = run_b2["bufs"]
    assert rel(b["encoder.image_encoder.features.bn1.running_mean"],
               gold["bn:encoder.image_encoder.features.bn1.running_mean"]) < 1e-4
    assert rel(b["encoder.image_encoder.features.bn1.running_var"],
               gold["bn:encoder.image_encoder.features.bn1.running_var"]) < 1e-4
    assert int(b["encoder.radar_encoder._model.layer3.1.bn2.num_batches_tracked"]) == 1
    worst = 0.0
    for k, v in run_b2["sdo"].items():
        if k.endswith(("running_mean", "running_var")):
            worst = max(worst, rel(b[k], v))
    assert worst < 1e-3, worst


def _oracle_fp64_grads(sd, rcfg, imgs, lids, rads, gps, target):
    """fp64 run of the oracle: the yardstick that calibrates how far two correct fp32 backward passes may differ
    (ReLU / max-pool decisions near zero make this gradient ill-conditioned in fp32)."""
    from oracle import fusion_ref as fr
    import torch.nn.functional as F
    dt = torch.float64
    sdo = {k: (v.to(dt).clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else
               (v.to(dt) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
    cast = lambda l: [t.to(dt) for t in l]
    lg = fr.transfuser_forward(sdo, cast(imgs), cast(lids), cast(rads), gps.to(dt), rcfg, fr.Ctx(training=True))
    t = target.double()
    p = torch.sigmoid(lg)
    ce = F.binary_cross_entropy_with_logits(lg, t, reduction="none")
    pt = p * t + (1 - p) * (1 - t)
    (((0.25 * t + 0.75 * (1 - t)) * ce * (1 - pt) ** 2).mean()).backward()
    return {k: v.grad for k, v in sdo.items() if isinstance(v, torch.Tensor) and v.requires_grad}


def test_gradients_match_oracle_and_golden(run_b2):
    """Parameter gradients of the HIP backward vs the CPU oracle.  Measured on this model: the fp32 CPU oracle
    itself deviates from an fp64 run by 1.2e-3 (median) / 7.8e-2 (max) of each tensor's largest entry, so the
    HIP path is held to the same yardstick: its deviation from fp64 must be within 3x of the fp32 oracle's,
    and no tensor may be off by more than 15 % (a wrong formula or a missing term gives O(1))."""
    from oracle import fusion_ref as fr
    gold = np.load(GOLD)
    r = run_b2
    rcfg = fr.RefConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 2, seed=100)
    g64 = _oracle_fp64_grads(fr.make_state(rcfg, seed=3), rcfg, imgs, lids, rads, gps, target)
    hip, o32 = [], []
    for k, ref in g64.items():
        scale = ref.abs().max().item()
        if scale < 1e-12:  # attn.key.bias: exactly-zero gradient (softmax is invariant to a per-query constant)
            assert r["grads"][k].abs().max().item() < 1e-7, k
            continue
        hip.append(((r["grads"][k].double() - ref).abs().max().item() / scale, k))
        o32.append(((r["sdo"][k].grad.double() - ref).abs().max().item() / scale, k))
    hip.sort(reverse=True)
    o32.sort(reverse=True)
    med = lambda v: v[len(v) // 2][0]
    print("grad err vs fp64: HIP median %.3e max %.3e (%s) | oracle32 median %.3e max %.3e" %
          (med(hip), hip[0][0], hip[0][1], med(o32), o32[0][0]), flush=True)
    assert med(hip) < 3 * med(o32) + 1e-4
    assert hip[0][0] < max(3 * o32[0][0], 0.05)
    assert hip[0][0] < 0.15, hip[0]
    # reference-generated fixture: gradient norms and leading entries
    for key in gold.files:
        if key.startswith("grad:") and key.endswith(":head"):
            name = key[len("grad:"):-len(":head")]
            g = r["grads"][name]
            head = g.flatten()[:16]  # .flatten() of the channels_last view walks the logical OIHW order
            scale = float(gold[f"grad:{name}:absmax"])
            e_head = (head - torch.from_numpy(gold[key])).abs().max().item() / scale
            e_norm = abs(float(g.norm()) - float(gold[f"grad:{name}:l2"])) / float(gold[f"grad:{name}:l2"])
            print(f"reference fixture {name}: head {e_head:.2e} of the largest entry, norm {e_norm:.2e}", flush=True)
            # per-tensor bars (round 4; 10 % / 5 % for every tensor before), ~10x what is measured on MI355X (head / norm):
            # join.4.bias 3.5e-7 / 8e-8, transformer4.blocks.7.mlp.2 8e-8 / 1e-7, vel_emb1 3e-6 / 3e-7, radar layer4 bn2 2e-5 /
            # 4e-7, transformer1.pos_emb 1e-3 / 3e-5, camera stem conv1 3e-3 / 3e-5 - the deeper the tensor sits below ReLU /
            # max-pool decisions, the larger single entries move; the norms stay at 1e-5-class everywhere
            bars = {"join.4.bias": (1e-5, 1e-5), "encoder.transformer4.blocks.7.mlp.2.weight": (1e-5, 1e-5),
                    "encoder.vel_emb1.weight": (1e-4, 1e-5), "encoder.radar_encoder._model.layer4.1.bn2.weight": (1e-3, 1e-4),
                    "encoder.transformer1.pos_emb": (1e-2, 1e-3), "encoder.image_encoder.features.conv1.weight": (3e-2, 1e-3)}
            bar_head, bar_norm = bars[name]
            assert e_head < bar_head and e_norm < bar_norm, (name, e_head, e_norm)


def test_autograd_boundary_and_grad_accumulation(dev):
    """loss.backward() through the nn.Module boundary gives the same grads as the fused harness path,
    and a second backward without zero_grad accumulates (torch semantics)."""
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    kw = dict(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, n_layer=1)
    model, rcfg, sd = _build(dev, kw, seed=5)
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 1, seed=7)
    model.train()
    loss1, _ = model.train_step_loss(imgs, lids, rads, gps, target)
    g1 = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    model.load_state_dict(sd)  # reset BN running stats
    logits = model(imgs, lids, rads, gps)
    loss2 = tr.sigmoid_focal_loss(logits, target.to(dev))
    loss2.backward()
    assert abs(float(loss1) - float(loss2)) < 1e-5
    for n, p in model.named_parameters():
        # absolute floor: attn.key.bias gradients are identically zero up to rounding noise (~1e-12)
        assert (p.grad - g1[n]).abs().max().item() < 1e-4 * g1[n].abs().max().item() + 1e-9, n
    # accumulate: second backward doubles the gradient
    logits = model(imgs, lids, rads, gps)
    tr.sigmoid_focal_loss(logits, target.to(dev)).backward()
    for n in ("join.4.weight", "encoder.transformer2.blocks.0.attn.query.weight",
              "encoder.image_encoder.features.layer2.0.downsample.0.weight",
              "encoder.lidar_encoder._model.conv1.weight", "encoder.transformer1.pos_emb",
              "encoder.radar_encoder._model.layer4.1.bn2.bias"):
        p = dict(model.named_parameters())[n]
        assert rel(p.grad, 2 * g1[n]) < 2e-3, n


def test_dropout_training_runs_and_is_seeded(dev):
    """Train mode with the reference's dropout (0.1): finite outputs, different masks on successive
    calls, and a loss in a sane range."""
    from oracle import fusion_ref as fr
    kw = dict(n_layer=2)
    model, rcfg, sd = _build(dev, kw, seed=11)
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 2, seed=3)
    model.train()
    l1, lg1 = model.train_step_loss(imgs, lids, rads, gps, target)
    l2, lg2 = model.train_step_loss(imgs, lids, rads, gps, target)
    assert torch.isfinite(lg1).all() and torch.isfinite(lg2).all()
    assert (lg1 - lg2).abs().max().item() > 1e-6
    for p in model.parameters():
        assert torch.isfinite(p.grad).all()


@pytest.mark.parametrize("seq_len,batch,n_layer", [(2, 3, 1), (10, 1, 1)])
def test_other_sequence_lengths_and_ragged_batch(dev, seq_len, batch, n_layer):
    """The path is generic in seq_len / batch: seq_len 2 with an odd batch (T = 386, 6 frames per trunk) and the
    reference's 30->5 sequence length 10 (T = 1922, model2_seq_30to5.py:624) against the oracle."""
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    kw = dict(seq_len=seq_len, n_layer=n_layer, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    model, rcfg, sd = _build(dev, kw, seed=21)
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, batch, seed=33)
    model.train()
    loss, logits = model.train_step_loss(imgs, lids, rads, gps, target)
    torch.set_num_threads(_host_threads())
    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    ref = fr.transfuser_forward(sdo, imgs, lids, rads, gps, rcfg, fr.Ctx(training=True))
    ref_loss = tr.sigmoid_focal_loss(ref, target)
    ref_loss.backward()
    assert rel(logits, ref.detach()) < TOL
    assert abs(float(loss) - float(ref_loss)) < TOL * abs(float(ref_loss))
    for name in ("join.4.weight", "encoder.transformer4.pos_emb", "encoder.vel_emb2.weight",
                 "encoder.image_encoder.features.layer4.2.conv2.weight"):
        # L2-relative: single entries of these gradients are ill-conditioned in fp32 (see the fp64-calibrated test)
        g = dict(model.named_parameters())[name].grad.cpu()
        assert ((g - sdo[name].grad).norm() / sdo[name].grad.norm()).item() < 5e-2, name


@pytest.mark.parametrize("kw_extra", [dict(n_views=2), dict(add_velocity=0)])
def test_config_variants_n_views_and_radar_channels(dev, kw_extra):
    """config.n_views = 2 (two camera frames per time step: pos_emb / token layout of model2_seq.py:189,261-263 with
    (n_views + 2) * seq_len * 64 + 2 tokens) and add_velocity = 0 (one radar channel, model2_seq.py:417-420)."""
    from oracle import fusion_ref as fr
    kw = dict(seq_len=2, n_layer=1, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, **kw_extra)
    model, rcfg, sd = _build(dev, kw, seed=23)
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 2, seed=34)
    assert len(imgs) == rcfg.n_views * rcfg.seq_len and rads[0].shape[1] == (2 if rcfg.add_velocity else 1)
    model.train()
    loss, logits = model.train_step_loss(imgs, lids, rads, gps, target)
    ref = fr.transfuser_forward(sd, imgs, lids, rads, gps, rcfg, fr.Ctx(training=True))
    assert rel(logits, ref) < TOL
    assert torch.isfinite(loss).all()


def test_fused_qkv_equals_separate_projections(dev):
    """The arena lays key|query|value out as one [3C, C] block and the model runs them as one GEMM; switching the
    fusion off (three GEMMs, the general path used when parameters were re-pointed) must give the same logits and
    gradients up to fp32 summation order, and the state-dict names/shapes are unaffected."""
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from oracle import fusion_ref as fr
    kw = dict(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0, n_layer=2)
    rcfg = fr.RefConfig(**kw)
    sd = fr.make_state(rcfg, seed=5)
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 2, seed=100)
    res = []
    for fuse in (True, False):
        model = TransFuser(GlobalConfig(**kw), dev)
        model.load_state_dict(sd)
        model.train()
        model.fuse_qkv = fuse
        at = model.encoder.transformer1.blocks[0].attn
        assert (model._qkv_fused(at) is not None)
        loss, logits = model.train_step_loss(imgs, lids, rads, gps, target)
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        res.append((logits.detach().clone(), grads))
        assert set(model.state_dict().keys()) == set(sd.keys())
    (l0, g0), (l1, g1) = res
    assert (l0 - l1).abs().max().item() <= 1e-4 * l1.abs().max().item()
    for n in ("encoder.transformer1.blocks.0.attn.key.weight", "encoder.transformer4.blocks.1.attn.query.bias",
              "encoder.transformer2.blocks.0.attn.value.weight", "encoder.image_encoder.features.conv1.weight"):
        d = (g0[n] - g1[n]).norm().item() / (g1[n].norm().item() + 1e-20)
        assert d < 5e-3, (n, d)


def test_eval_bn_folding_matches_unfolded(dev):
    """model.eval(): BatchNorm folded into the conv weights (conv + bias / ReLU / identity epilogues, no BN kernels)
    must equal the unfolded eval path (bn_eval_prepare + bn_apply) to fp32 rounding, after running statistics have
    moved away from their initial values."""
    from oracle import fusion_ref as fr
    kw = dict(n_layer=1, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    model, rcfg, sd = _build(dev, kw, seed=31)
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 2, seed=35)
    model.train()
    model.train_step_loss(imgs, lids, rads, gps, target)   # updates the BN running statistics
    model.eval()
    with torch.no_grad():
        model.fold_bn_eval = True
        a = model(imgs, lids, rads, gps)
        model.fold_bn_eval = False
        b = model(imgs, lids, rads, gps)
    assert rel(a, b) < 2e-5
    ref = fr.transfuser_forward({k: v.cpu() for k, v in model.state_dict().items()}, imgs, lids, rads, gps, rcfg,
                                fr.Ctx(training=False))
    assert rel(a, ref) < TOL


def test_captured_inference_graph_matches_eager(dev):
    """capture_inference(): the eval forward replayed from one HIP graph equals the eager forward bit for bit, follows
    new inputs (static input buffers) and new weights (parameters are read at replay time)."""
    from oracle import fusion_ref as fr
    kw = dict(n_layer=1, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    model, rcfg, sd = _build(dev, kw, seed=41)
    model.eval()
    a = fr.make_inputs(rcfg, 1, seed=50)[:4]
    b = fr.make_inputs(rcfg, 1, seed=51)[:4]
    run = model.capture_inference(*a)
    with torch.no_grad():
        assert torch.equal(run(*a), model(*a))
        assert torch.equal(run(*b), model(*b))
        model.join[4].bias.data.add_(1.0)          # in-place weight change is seen by the next replay
        assert torch.equal(run(*b), model(*b))


def test_gpt_stage_alone_matches_reference_fixture_and_oracle_gradients(dev):
    """One GPT fusion stage in isolation: the token pack / pos_emb / 2 Blocks / ln_f / unpack of
    model2_seq.py:248-287 on the HIP kernels against the REFERENCE-generated fixture `gpt1_*` (tests/golden/make_golden.py ran
    the reference's own GPT on these inputs), and every parameter gradient of that stage against the oracle's autograd at a
    tight per-tensor bar (1e-4 of the tensor's largest entry: this sub-graph has no BatchNorm / max-pool decisions, so a
    mis-scaled LayerNorm beta, bias or pos_emb gradient cannot hide behind conditioning)."""
    from deepsense6g_tii_amd import ops
    from deepsense6g_tii_amd._lib import lib
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from oracle import fusion_ref as fr
    gold = np.load(GOLD)
    kw = dict(seq_len=1, n_layer=2, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    cfg1 = fr.RefConfig(**kw)
    sd1 = fr.make_state(cfg1, seed=7)
    g = torch.Generator().manual_seed(11)
    img, lid, rad = (torch.randn(2, 64, 8, 8, generator=g) for _ in range(3))
    gps = torch.randn(2, 2, 64, generator=g)
    model = TransFuser(GlobalConfig(**kw), dev)
    model.load_state_dict(sd1)
    model.train()
    L, st = lib(), ops._stream()
    B, S, C, T = 2, 1, 64, 3 * 64 + 2
    gpt = model.encoder.transformer1
    feats = [t.permute(0, 2, 3, 1).contiguous().to(dev) for t in (img, lid, rad)]
    x0 = torch.empty((B, T, C), device=dev)
    pos = gpt.pos_emb.data_ptr()
    for m in range(3):
        L.avgpool_tokens_fwd(feats[m].data_ptr(), pos, x0.data_ptr(), B * S, 8, C, S, m * S * 64, T, 0.0, 0, 0, st)
    gemb = gps.to(dev).contiguous()
    L.gps_tokens_fwd(gemb.data_ptr(), pos, x0.data_ptr(), B, C, T, 0.0, 0, 0, st)
    model._recording, model._use16 = True, False
    x, ctxs = x0.view(B * T, C), []
    for blk in gpt.blocks:
        x, c = model._gpt_block_fwd(blk, x, B, T, True)
        ctxs.append(c)
    xo, mf, rf = ops.layernorm_fwd(x, gpt.ln_f.weight.data_ptr(), gpt.ln_f.bias.data_ptr(), gpt.ln_f.eps)
    out = xo.view(B, T, C).cpu()
    for m, key in enumerate(("gpt1_img", "gpt1_lid", "gpt1_rad")):
        maps = out[:, m * 64:(m + 1) * 64].reshape(B, 8, 8, C).permute(0, 3, 1, 2)     # token row t*64 + h*8 + w -> NCHW
        assert np.abs(maps.numpy() - gold[key]).max() < 1e-5, key
    assert np.abs(out[:, T - 2:].numpy() - gold["gpt1_gps"]).max() < 1e-5
    # ---- backward of the stage against the oracle's autograd, upstream = a fixed random cotangent on every token
    up = torch.randn(B, T, C, generator=g)
    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd1.items()}
    o = fr.gpt_stage(sdo, "encoder.transformer1.", img, lid, rad, gps, cfg1, fr.Ctx(training=True))
    flat = torch.cat([t.reshape(B, C, 64).permute(0, 2, 1) for t in o[:3]] + [o[3]], dim=1)   # (B, T, C) in token order
    (flat * up).sum().backward()
    for p in model.parameters():
        p.grad = None
    model._begin_backward()
    gw, aw = model._g(gpt.ln_f.weight)
    gb, _ = model._g(gpt.ln_f.bias)
    dx = ops.layernorm_bwd(up.view(B * T, C).to(dev).contiguous(), x, mf, rf, gpt.ln_f.weight.data_ptr(), gw, gb, model._ws,
                           accumulate=bool(aw))
    for blk, c in zip(reversed(list(gpt.blocks)), reversed(ctxs)):
        dx, _ = model._gpt_block_bwd(blk, c, dx, B, T)
    model._wg_join()
    gpos, apos = model._g(gpt.pos_emb)
    L.batch_sum(dx.data_ptr(), gpos, T * C, B, T * C, apos, st)
    model._end_backward()
    torch.cuda.synchronize()
    worst = []
    top = max(sdo["encoder.transformer1." + n].grad.abs().max().item() for n, _ in gpt.named_parameters())
    for name, p in gpt.named_parameters():
        ref = sdo["encoder.transformer1." + name].grad
        scale = ref.abs().max().item()
        if scale < 1e-6 * top:  # attn.key.bias: zero up to rounding noise (softmax is invariant to a per-query constant)
            assert name.endswith("attn.key.bias") and p.grad.abs().max().item() < 1e-6 * top, name
            continue
        worst.append(((p.grad.cpu() - ref).abs().max().item() / scale, name))
    worst.sort(reverse=True)
    print("GPT-stage gradient errors (worst 3):", worst[:3])
    assert worst[0][0] < 1e-4, worst[:5]


@pytest.mark.parametrize("layer,idx", [("layer2", 0), ("layer1", 1)])
def test_basic_block_alone_gradients_match_torch_autograd(dev, layer, idx):
    """One BasicBlock in isolation (with and without the strided 1x1 downsample branch), train-mode BatchNorm, forward and
    every gradient (conv weights, BN gamma / beta of all three norms, input) against torch autograd on the CPU at 1e-4 of
    each tensor's largest entry (the whole-model gradient bars are necessarily loose - ill-conditioned
    ReLU / max-pool decisions over 34 layers - so a mis-scaled BN beta or downsample gradient is pinned here instead)."""
    import torch.nn.functional as F
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    kw = dict(n_layer=1, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    torch.manual_seed(5)
    model = TransFuser(GlobalConfig(**kw), dev)
    model.train()
    blk = getattr(model.encoder.image_encoder.features, layer)[idx]
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():   # non-trivial BN affine parameters
        for bn in [blk.bn1, blk.bn2] + ([blk.downsample[1]] if blk.downsample is not None else []):
            bn.weight.copy_((torch.rand(bn.weight.shape, generator=g) + 0.5).to(dev))
            bn.bias.copy_((torch.randn(bn.bias.shape, generator=g) * 0.3).to(dev))
    Cin = blk.conv1.in_channels
    x = torch.randn(4, Cin, 16, 16, generator=g)
    # ---- torch reference on the CPU
    ref = {n: p.detach().cpu().clone().contiguous().requires_grad_(True) for n, p in blk.named_parameters()}
    xr = x.clone().requires_grad_(True)

    def bn(t, pre):
        return F.batch_norm(t, None, None, ref[pre + ".weight"], ref[pre + ".bias"], training=True, eps=1e-5)
    o = F.relu(bn(F.conv2d(xr, ref["conv1.weight"], None, blk.stride, 1), "bn1"))
    o = bn(F.conv2d(o, ref["conv2.weight"], None, 1, 1), "bn2")
    idn = bn(F.conv2d(xr, ref["downsample.0.weight"], None, blk.stride, 0), "downsample.1") if blk.downsample is not None else xr
    y = F.relu(o + idn)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    # ---- HIP kernels through the model's own block walk
    model._recording, model._use16, model._fold_now = True, False, False
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    yg, ctx = model._block_fwd(blk, xg, True)
    assert rel(yg.cpu().permute(0, 3, 1, 2), y.detach()) < 2e-5
    for p in model.parameters():
        p.grad = None
    model._begin_backward()
    dxg = model._block_bwd(blk, ctx, dy.permute(0, 2, 3, 1).contiguous().to(dev))
    model._wg_join()
    model._end_backward()
    torch.cuda.synchronize()
    errs = [(rel(dxg.cpu().permute(0, 3, 1, 2), xr.grad), "dx")]
    for n, p in blk.named_parameters():
        errs.append((rel(p.grad.cpu(), ref[n].grad), n))
    errs.sort(reverse=True)
    print("BasicBlock gradient errors:", errs[:4])
    assert errs[0][0] < 1e-4, errs
