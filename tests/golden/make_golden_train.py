"""Generates tests/golden/train_golden.npz by running the REFERENCE's own plain-Python training pieces (build
container only; /root/reference never travels).

/root/reference/train2_seq.py cannot be imported: it parses argv, opens a SummaryWriter and builds the dataset at
import time (:30-61, :70, :457+) and needs tensorboard / open3d / utm / cv2 / mamba_ssm.  The functions this path
needs are pure numpy / torch, so this script parses the file with `ast`, picks the definitions BY NAME -
`EMA` (:303-334), `save_pred_to_csv` (:338-346), `compute_acc` (:347-360), `compute_DBA_score` (:363-383) - and
executes exactly those definitions in a namespace holding numpy / pandas / torch.  `scheduler.py` imports, but
`CyclicCosineDecayLR.__init__` passes `verbose=` to `_LRScheduler.__init__` (:80), which torch >= 2.7 no longer
accepts: the class definition is executed with `_LRScheduler` bound to a shim base class that swallows that one
keyword - every line of the schedule itself (get_lr :82-115, _calc :117-119) is the reference's.

Stored: seeded inputs + the reference's outputs (numbers and the csv text it wrote) - no reference source text.
  lr_base{1e-4,5e-4}   lr of epochs 0..60 with the arguments of train2_seq.py:541-547, stepped as :613-615
  acc_* / dba_*        compute_acc / compute_DBA_score on random argsorts (several n, max_k, delta)
  ema_*                EMA.register -> 5 x (perturb params, update) -> shadow; apply_shadow / restore swap
  csv_text             beam_pred.csv written by save_pred_to_csv
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py
"""
import ast
import io
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import pandas as pd
import torch
from torch import nn

REF = "/root/reference"


def _extract(path, names):
    """-> namespace-ready code object holding only the top-level definitions `names` of the file at `path`"""
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), [n.name for n in picked]
    return compile(ast.Module(body=picked, type_ignores=[]), path, "exec")


def reference_train_pieces():
    ns = dict(np=np, pd=pd, torch=torch)
    exec(_extract(os.path.join(REF, "train2_seq.py"), ["EMA", "save_pred_to_csv", "compute_acc", "compute_DBA_score"]), ns)
    return ns


def reference_scheduler():
    from collections.abc import Iterable
    from math import cos, floor, log, pi

    from torch.optim.lr_scheduler import _LRScheduler

    class _Shim(_LRScheduler):  # drops the `verbose=` keyword torch >= 2.7 removed; nothing else
        def __init__(self, optimizer, last_epoch=-1, verbose=False):
            super().__init__(optimizer, last_epoch)

    ns = dict(Iterable=Iterable, cos=cos, floor=floor, log=log, pi=pi, _LRScheduler=_Shim)
    exec(_extract(os.path.join(REF, "scheduler.py"), ["CyclicCosineDecayLR"]), ns)
    return ns["CyclicCosineDecayLR"]


def tiny_model(seed):
    torch.manual_seed(seed)
    m = nn.Sequential(nn.Linear(7, 5), nn.ReLU(), nn.Linear(5, 3))
    m[2].bias.requires_grad_(False)  # EMA skips parameters that do not require grad (train2_seq.py:312)
    return m


def main():
    out = {}
    ns = reference_train_pieces()
    Sched = reference_scheduler()

    # ---- schedule: constructed and stepped exactly as train2_seq.py:541-547, :613-615 ----
    import warnings
    for base in (1e-4, 5e-4):
        p = nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=base)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sch = Sched(opt, init_decay_epochs=15, min_decay_lr=2.5e-6, restart_interval=10, restart_lr=12.5e-5,
                        warmup_epochs=10, warmup_start_lr=2.5e-6)
            lrs = []
            for epoch in range(61):
                lrs.append(opt.param_groups[0]["lr"])   # the lr the epoch trains with
                opt.step()
                sch.step()
        out[f"lr_base{base:g}"] = np.asarray(lrs, dtype=np.float64)

    # ---- metrics on random argsorts ----
    rng = np.random.default_rng(20240917)
    for i, n in enumerate((1, 7, 64, 500)):
        scores = rng.standard_normal((n, 64))
        pred = np.argsort(-scores, axis=1)
        true = rng.integers(0, 64, size=n)
        if n >= 64:   # make part of the labels hits / near misses so that every branch of min(d/delta, 1) is visited
            true[: n // 3] = pred[: n // 3, 0]
            true[n // 3: n // 2] = np.clip(pred[n // 3: n // 2, 1] + rng.integers(-6, 7, size=n // 2 - n // 3), 0, 63)
        out[f"metric{i}_pred"], out[f"metric{i}_true"] = pred.astype(np.int64), true.astype(np.int64)
        out[f"metric{i}_acc"] = ns["compute_acc"](pred, true, top_k=[1, 2, 3])
        out[f"metric{i}_acc5"] = ns["compute_acc"](pred, true, top_k=[1, 3, 5])
        out[f"metric{i}_dba"] = np.float64(ns["compute_DBA_score"](pred, true, max_k=3, delta=5))
        out[f"metric{i}_dba_k5_d3"] = np.float64(ns["compute_DBA_score"](pred, true, max_k=5, delta=3))

    # ---- EMA: register, 5 updates, shadow swap ----
    m = tiny_model(7)
    ema = ns["EMA"](m, 0.999)
    ema.register()
    g = torch.Generator().manual_seed(11)
    steps = []
    for _ in range(5):
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        ema.update()
        steps.append(torch.cat([p.detach().flatten() for p in m.parameters()]).numpy().copy())
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    out["ema_names"] = np.asarray(names)
    out["ema_param_steps"] = np.stack(steps)                                   # all parameters after each perturbation
    out["ema_shadow"] = torch.cat([ema.shadow[n].flatten() for n in names]).numpy()
    live = {n: p.data for n, p in m.named_parameters()}
    ema.apply_shadow()
    out["ema_applied_is_shadow"] = np.asarray([bool(p.data.data_ptr() == ema.shadow[n].data_ptr())
                                               for n, p in m.named_parameters() if p.requires_grad])
    ema.restore()
    out["ema_restored_is_live"] = np.asarray([bool(p.data.data_ptr() == live[n].data_ptr()) for n, p in m.named_parameters()])
    assert out["ema_applied_is_shadow"].all() and out["ema_restored_is_live"].all() and ema.backup == {}

    # ---- csv ----
    pred = out["metric1_pred"]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "beam_pred.csv")
        ns["save_pred_to_csv"](pred, target_csv=path)
        out["csv_text"] = np.asarray(open(path).read())
        path5 = os.path.join(d, "beam_pred5.csv")
        ns["save_pred_to_csv"](pred, top_k=[1, 2, 3, 4, 5], target_csv=path5)
        out["csv_text_top5"] = np.asarray(open(path5).read())

    # ---- the oracle restatement against what the reference just produced (the committed report) ----
    from oracle import train_ref as tr
    lines = []
    for base in (1e-4, 5e-4):
        d = max(abs(tr.cyclic_cosine_lr(e, base) - out[f"lr_base{base:g}"][e]) for e in range(61))
        lines.append(f"cyclic_cosine_lr base {base:g}: max |oracle - reference| over epochs 0..60 = {d:.3e}")
    for i in range(4):
        pred, true = out[f"metric{i}_pred"], out[f"metric{i}_true"]
        lines.append(f"metrics n={len(true)}: acc diff {np.abs(tr.compute_acc(pred, true) - out[f'metric{i}_acc']).max():.1e}, "
                     f"DBA diff {abs(tr.compute_dba_score(pred, true) - float(out[f'metric{i}_dba'])):.1e}")
    print("\n".join(lines))
    np.savez_compressed(os.path.join(HERE, "train_golden.npz"), **out)
    with open(os.path.join(HERE, "oracle_vs_reference_train.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote train_golden.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
