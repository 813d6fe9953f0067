"""CPU ORACLE (test infrastructure, never the product path) for the training-step pieces around the
fusion model: sigmoid focal loss, AdamW, EMA, the cyclic-cosine LR schedule and the DBA / top-k
metrics.  /root/reference/train2_seq.py cannot be imported (argparse + SummaryWriter + dataset at
import time, :61,:70,:457+) so each function restates the source text it cites.

Parity status: `cyclic_cosine_lr`, `compute_acc`, `compute_dba_score` and `ema_update` are PINNED to outputs of the
reference's own functions (tests/golden/train_golden.npz, made in the build container by
tests/golden/make_golden_train.py, which `ast`-extracts and executes train2_seq.py:303-383 and scheduler.py:7-119;
bit-equal, tests/test_oracle_cpu.py::test_*_match(es)_reference_run) on top of the hand-derived known answers there.
`adamw_step` is pinned to torch.optim.AdamW (tests/test_oracle_cpu.py::test_adamw_and_ema_match_torch).
torchvision.ops.sigmoid_focal_loss is an un-vendored dependency (version unpinned, not importable here): restated from
its published formula -> "parity unpinned" at that boundary.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def sigmoid_focal_loss(logits: torch.Tensor, target: torch.Tensor, alpha: float = 0.25,
                       gamma: float = 2.0) -> torch.Tensor:
    """train2_seq.py:291-301 -> torchvision.ops.sigmoid_focal_loss(..., reduction='mean').

    p = sigmoid(x); ce = BCEWithLogits(x, t); p_t = p t + (1-p)(1-t);
    loss = ce (1-p_t)^gamma (alpha t + (1-alpha)(1-t)); mean over all B*64 elements.
    A 1-D integer target is one-hot encoded to 64 classes first (:297-298).
    """
    if target.dim() == 1:
        target = F.one_hot(target.long(), num_classes=64)
    target = target.float()
    p = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(logits, target, reduction="none")
    p_t = p * target + (1 - p) * (1 - target)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * target + (1 - alpha) * (1 - target)) * loss
    return loss.mean()


def adamw_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW defaults as used at train2_seq.py:539 (decoupled wd 0.01 on ALL params).
    In-place on p, m, v; ``step`` is the 1-based step count."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def ema_update(shadow, p, decay=0.999):
    """train2_seq.py:315-320: shadow = (1-d) p + d shadow."""
    return (1.0 - decay) * p + decay * shadow


def cyclic_cosine_lr(epoch: int, base_lr: float, init_decay_epochs=15, min_decay_lr=2.5e-6,
                     restart_interval=10, restart_lr=12.5e-5, warmup_epochs=10,
                     warmup_start_lr=2.5e-6) -> float:
    """scheduler.py:82-119 with the arguments of train2_seq.py:541-547
    (restart_interval_multiplier is None there)."""

    def calc(t, T, lr, min_lr):  # scheduler.py:117-119
        return min_lr + (lr - min_lr) * ((1 + math.cos(math.pi * t / T)) / 2)

    if warmup_epochs > 0 and epoch < warmup_epochs:
        return calc(epoch, warmup_epochs, warmup_start_lr, base_lr)
    if epoch < init_decay_epochs + warmup_epochs:
        return calc(epoch - warmup_epochs, init_decay_epochs, base_lr, min_decay_lr)
    if restart_interval is None:
        return min_decay_lr
    cyc = (epoch - init_decay_epochs - warmup_epochs) % restart_interval
    return calc(cyc, restart_interval, base_lr if restart_lr is None else restart_lr, min_decay_lr)


def compute_acc(y_pred: np.ndarray, y_true: np.ndarray, top_k=(1, 2, 3)) -> np.ndarray:
    """train2_seq.py:347-360. y_pred: (n,64) argsorted beams (descending score)."""
    if len(y_pred) != len(y_true):
        raise Exception("Number of predicted beams does not match number of labels.")
    hits = np.zeros(len(top_k))
    for i in range(len(y_true)):
        for j, k in enumerate(top_k):
            hits[j] += 1 if np.any(y_pred[i, :k] == y_true[i]) else 0
    return np.round(hits / len(y_true) * 100, 4)


def compute_dba_score(y_pred: np.ndarray, y_true: np.ndarray, max_k=3, delta=5) -> float:
    """train2_seq.py:363-383."""
    n = y_pred.shape[0]
    yk = np.zeros(max_k)
    for k in range(max_k):
        acc = 0.0
        for i in range(n):
            d = np.abs(y_pred[i, : k + 1] - y_true[i]) / delta
            acc += np.min(np.minimum(d, 1.0))
        yk[k] = 1 - acc / n
    return float(np.mean(yk))
