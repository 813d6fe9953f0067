"""CPU: the oracle (oracle/fusion_ref.py) against the golden fixture produced by the REFERENCE code
(tests/golden/fusion_golden.npz, generator tests/golden/make_golden.py), plus known-answer tests of the
restated training pieces (oracle/train_ref.py).  No GPU, no /root/reference at run time."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import fusion_ref as fr
from oracle import train_ref as tr

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fusion_golden.npz")


def test_param_table_matches_reference_contract():
    cfg = fr.RefConfig()
    shapes = fr.param_shapes(cfg)
    n_params = sum(int(np.prod(s)) for k, s in shapes.items() if not fr.is_buffer(k))
    assert n_params == 78_422_528  # SURVEY.md 2.2 (reference model, GPT variant, add_velocity=1)
    assert shapes["encoder.transformer4.pos_emb"] == (1, 962, 512)
    assert shapes["encoder.radar_encoder._model.conv1.weight"] == (64, 2, 7, 7)
    assert shapes["encoder.lidar_encoder._model.conv1.weight"] == (64, 1, 7, 7)
    assert shapes["encoder.image_encoder.features.layer3.0.downsample.0.weight"] == (256, 128, 1, 1)
    assert "encoder.image_encoder.features.layer1.0.downsample.0.weight" not in shapes


def test_gpt_stage_matches_reference_fixture():
    gold = np.load(GOLD)
    cfg1 = fr.RefConfig(seq_len=1, n_layer=2, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    sd1 = fr.make_state(cfg1, seed=7)
    g = torch.Generator().manual_seed(11)
    img, lid, rad = (torch.randn(2, 64, 8, 8, generator=g) for _ in range(3))
    gps = torch.randn(2, 2, 64, generator=g)
    with torch.no_grad():
        o = fr.gpt_stage(sd1, "encoder.transformer1.", img, lid, rad, gps, cfg1, fr.Ctx(training=True))
    for key, t in zip(("gpt1_img", "gpt1_lid", "gpt1_rad", "gpt1_gps"), o):
        assert np.abs(t.numpy() - gold[key]).max() < 1e-5, key


@pytest.fixture(scope="module")
def oracle_b2():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    cfg = fr.RefConfig(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    sd = fr.make_state(cfg, seed=3)
    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    imgs, lids, rads, gps, target, _ = fr.make_inputs(cfg, 2, seed=100)
    cap = {}
    logits = fr.transfuser_forward(sdo, imgs, lids, rads, gps, cfg, fr.Ctx(training=True, capture=cap))
    loss = tr.sigmoid_focal_loss(logits, target)
    loss.backward()
    with torch.no_grad():
        logits_eval = fr.transfuser_forward(sdo, imgs, lids, rads, gps, cfg, fr.Ctx(training=False))
    return sdo, logits.detach(), cap, loss.detach(), logits_eval


def test_full_path_matches_reference_fixture(oracle_b2):
    gold = np.load(GOLD)
    sdo, logits, cap, loss, logits_eval = oracle_b2
    assert np.abs(logits.numpy() - gold["logits_b2"]).max() < 2e-5
    assert np.abs(cap["fused"].detach().numpy() - gold["fused_b2"]).max() < 2e-4
    assert abs(float(loss) - float(gold["loss_b2"])) < 1e-6
    assert np.abs(logits_eval.numpy() - gold["logits_eval_b2"]).max() < 2e-5
    for name in ("running_mean", "running_var"):
        k = "encoder.image_encoder.features.bn1." + name
        assert np.abs(sdo[k].numpy() - gold["bn:" + k]).max() < 1e-5
    assert int(sdo["encoder.image_encoder.features.bn1.num_batches_tracked"]) == 1


def test_gradients_match_reference_fixture(oracle_b2):
    gold = np.load(GOLD)
    sdo = oracle_b2[0]
    for key in gold.files:
        if key.startswith("grad:") and key.endswith(":head"):
            name = key[len("grad:"):-len(":head")]
            g = sdo[name].grad
            scale = float(gold[f"grad:{name}:absmax"])
            assert np.abs(g.flatten()[:16].numpy() - gold[key]).max() < 2e-3 * scale + 1e-9, name
            assert abs(float(g.norm()) - float(gold[f"grad:{name}:l2"])) < 2e-3 * float(gold[f"grad:{name}:l2"])


# ---------------------------------------------------------------- train_ref known answers -------
def test_focal_loss_known_answers():
    # x = 0, t = 0: p = .5, ce = ln2, (1-p_t)^2 = .25, alpha_t = .75  -> .75*.25*ln2
    x = torch.zeros(3, 64)
    assert abs(float(tr.sigmoid_focal_loss(x, torch.zeros(3, 64))) - 0.75 * 0.25 * math.log(2)) < 1e-7
    # t = 1 everywhere: alpha_t = .25
    assert abs(float(tr.sigmoid_focal_loss(x, torch.ones(3, 64))) - 0.25 * 0.25 * math.log(2)) < 1e-7
    # integer targets are one-hot encoded (train2_seq.py:297-298)
    idx = torch.tensor([0, 5, 63])
    one_hot = torch.nn.functional.one_hot(idx, 64).float()
    xr = torch.randn(3, 64, generator=torch.Generator().manual_seed(0))
    assert float(tr.sigmoid_focal_loss(xr, idx)) == float(tr.sigmoid_focal_loss(xr, one_hot))
    # confident and right -> ~0 ; confident and wrong -> large
    big = torch.full((1, 64), -20.0)
    assert float(tr.sigmoid_focal_loss(big, torch.zeros(1, 64))) < 1e-12
    assert float(tr.sigmoid_focal_loss(-big, torch.zeros(1, 64))) > 10.0


def test_adamw_and_ema_match_torch():
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(257, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-3)  # defaults = train2_seq.py:539
    p, m, v = p0.clone(), torch.zeros(257), torch.zeros(257)
    shadow = p0.clone()
    hist = []
    for step in range(1, 5):
        grad = torch.randn(257, generator=g)
        ref.grad = grad.clone()
        opt.step()
        tr.adamw_step(p, grad, m, v, step, 1e-3)
        shadow = tr.ema_update(shadow, p, 0.999)
        hist.append(p.clone())
    assert (p - ref.detach()).abs().max() < 1e-6
    # closed form of train2_seq.py:315-320: shadow_n = d^n p0 + (1-d) sum_i d^(n-i) p_i
    d = 0.999
    closed = d ** 4 * p0 + (1 - d) * sum(d ** (4 - i) * hist[i - 1] for i in range(1, 5))
    assert (shadow - closed).abs().max() < 1e-6


def test_cyclic_cosine_schedule_known_answers():
    # hand-derived from scheduler.py:84-103 with the arguments of train2_seq.py:541-547
    base = 1e-4
    assert abs(tr.cyclic_cosine_lr(0, base) - 2.5e-6) < 1e-12        # warm-up start
    assert abs(tr.cyclic_cosine_lr(10, base) - base) < 1e-12           # warm-up end = base lr
    assert abs(tr.cyclic_cosine_lr(25, base) - 1.25e-4) < 1e-12        # first restart
    assert abs(tr.cyclic_cosine_lr(35, base) - 1.25e-4) < 1e-12        # second restart
    mid = 2.5e-6 + (1.25e-4 - 2.5e-6) * 0.5
    assert abs(tr.cyclic_cosine_lr(30, base) - mid) < 1e-12            # half-way through a cycle
    lrs = [tr.cyclic_cosine_lr(e, base) for e in range(10, 25)]
    assert all(a > b for a, b in zip(lrs, lrs[1:]))                    # monotone decay 10..24


def test_metrics_known_answers():
    y = np.array([3, 10, 63])
    perfect = np.stack([np.roll(np.arange(64), -t) for t in y])       # first column == truth
    assert tr.compute_dba_score(perfect, y) == 1.0
    assert list(tr.compute_acc(perfect, y)) == [100.0, 100.0, 100.0]
    far = np.stack([np.roll(np.arange(64), -((t + 20) % 64)) for t in y])  # top-3 all >= 5 beams away
    assert tr.compute_dba_score(far, y) == 0.0
    assert list(tr.compute_acc(far, y)) == [0.0, 0.0, 0.0]
    # one beam off in top-1, exact in top-2: k=1 -> 1-1/5, k=2,3 -> 1
    pred = np.array([[4, 3, 50] + [0] * 61])
    assert abs(tr.compute_dba_score(pred, np.array([3])) - (0.8 + 1 + 1) / 3) < 1e-12
    assert list(tr.compute_acc(pred, np.array([3]))) == [0.0, 100.0, 100.0]
    with pytest.raises(Exception):
        tr.compute_acc(pred, np.array([1, 2]))


# ---- the restated training pieces against outputs of the REFERENCE's own functions (tests/golden/train_golden.npz, made
# by tests/golden/make_golden_train.py from train2_seq.py:303-383 and scheduler.py:7-119 in the build container) ----
TRAIN_GOLD = os.path.join(os.path.dirname(__file__), "golden", "train_golden.npz")


def test_schedule_matches_reference_run():
    g = np.load(TRAIN_GOLD)
    for base in (1e-4, 5e-4):
        ref = g[f"lr_base{base:g}"]
        assert ref.shape == (61,)
        for e in range(61):
            assert tr.cyclic_cosine_lr(e, base) == ref[e], (base, e)     # same arithmetic, same order: bit-equal


def test_metrics_match_reference_run():
    g = np.load(TRAIN_GOLD)
    for i in range(4):
        pred, true = g[f"metric{i}_pred"], g[f"metric{i}_true"]
        assert np.array_equal(tr.compute_acc(pred, true), g[f"metric{i}_acc"])
        assert np.array_equal(tr.compute_acc(pred, true, top_k=(1, 3, 5)), g[f"metric{i}_acc5"])
        assert tr.compute_dba_score(pred, true) == float(g[f"metric{i}_dba"])
        assert tr.compute_dba_score(pred, true, max_k=5, delta=3) == float(g[f"metric{i}_dba_k5_d3"])


def test_ema_matches_reference_run():
    """EMA.register -> 5 x (perturb, update) of the reference class on a tiny module (one parameter frozen: the reference
    skips requires_grad = False); the oracle's ema_update on the same parameter sequence must give the same shadow."""
    g = np.load(TRAIN_GOLD)
    steps = torch.from_numpy(g["ema_param_steps"])            # [5, 58]: all parameters after each perturbation
    n_tracked = g["ema_shadow"].shape[0]                      # 55: the frozen bias (3 values, last) is not tracked
    assert steps.shape == (5, 58) and n_tracked == 55 and list(g["ema_names"]) == ["0.weight", "0.bias", "2.weight"]
    from tests.golden.make_golden_train import tiny_model
    m = tiny_model(7)
    shadow = torch.cat([p.detach().flatten() for p in m.parameters()])[:n_tracked].clone()     # register()
    for s in range(5):
        shadow = tr.ema_update(shadow, steps[s, :n_tracked], 0.999)
    assert torch.equal(shadow, torch.from_numpy(g["ema_shadow"]))


# ---- the 30 -> 5 variant (seq_len 10 => 1922 tokens, GRU head): fixture made by the REFERENCE's own model2_seq_30to5.py
# Encoder + TransFuser.forward (tests/golden/make_golden_30to5.py) ---------------------------------------------------------
GOLD30 = os.path.join(os.path.dirname(__file__), "golden", "fusion30to5_golden.npz")


def golden30_case():
    """-> (RefConfig, state dict, inputs, target) of the fixture, rebuilt from its stored seeds"""
    gold = np.load(GOLD30)
    state_seed, input_seed, target_seed, batch, seq_len, n_layer, pred_len = (int(v) for v in gold["meta"])
    cfg = fr.RefConfig(seq_len=seq_len, n_layer=n_layer, pred_len=pred_len, gru_head=True, embd_pdrop=0.0, attn_pdrop=0.0,
                       resid_pdrop=0.0)
    sd = fr.make_state(cfg, seed=state_seed)
    imgs, lids, rads, gps, _, _ = fr.make_inputs(cfg, batch, seed=input_seed)
    target = torch.rand(batch, pred_len, 64, generator=torch.Generator().manual_seed(target_seed)) * 0.5
    return gold, cfg, sd, (imgs, lids, rads, gps), target


def test_30to5_variant_matches_reference_fixture():
    """oracle at seq_len 10 (pos_emb of 1922 tokens, model2_seq_30to5.py:188) + join + GRUCell / Linear head unrolled
    pred_len = 5 times (:846-862) against the outputs of the reference's own classes: predictions, fused features, loss,
    nine gradient probes from the head down to the camera stem."""
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    gold, cfg, sd, inputs, target = golden30_case()
    assert cfg.n_tokens == 1922
    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    cap = {}
    pred = fr.transfuser_forward(sdo, *inputs, cfg, fr.Ctx(training=True, capture=cap))
    loss = tr.sigmoid_focal_loss(pred, target)
    loss.backward()
    assert tuple(pred.shape) == (2, 5, 64)
    assert np.abs(pred.detach().numpy() - gold["pred_b2"]).max() < 2e-5
    assert np.abs(cap["fused"].detach().numpy() - gold["fused_b2"]).max() < 2e-4
    assert abs(float(loss) - float(gold["loss_b2"])) < 1e-6
    probes = [k[len("grad:"):-len(":head")] for k in gold.files if k.startswith("grad:") and k.endswith(":head")]
    assert len(probes) == 9
    for name in probes:
        g = sdo[name].grad.flatten()
        scale = float(gold[f"grad:{name}:absmax"])
        assert np.abs(g[:16].numpy() - gold[f"grad:{name}:head"]).max() < 2e-4 * scale + 1e-12, name
        assert np.abs(g[::max(1, g.numel() // 64)][:64].numpy() - gold[f"grad:{name}:strided"]).max() < 2e-4 * scale + 1e-12, name
        assert abs(float(g.norm()) - float(gold[f"grad:{name}:l2"])) < 1e-4 * float(gold[f"grad:{name}:l2"]), name
