"""CPU ORACLE run of the synthetic *learnable* beam task (test infrastructure, never the product path): the reference-side
half of the "DBA top-k on held-out" metric ("identically for the CPU reference path and the MI355X build").

Same protocol as tools/train_synthetic.py (the HIP run): same initial weights (torch.manual_seed(100) + the TransFuser
parameter containers' init), same training batches (seeds 100 + i), same held-out seeds (10 000 + i), AdamW lr 1e-4 wd 0.01,
sigmoid focal loss on the soft target, dropout 0.1 (torch's own Philox masks - the HIP path draws other masks of the same
distribution), train-mode BatchNorm; evaluation = eval-mode forward -> argsort -> DBA / top-k (train2_seq.py:158-221,
347-383).  Writes one JSON line per evaluation, like the HIP tool.

    python oracle/train_synthetic_cpu.py --steps 50 --eval-every 25 --threads 6 > profiles/r02_train_synthetic_dba_cpu.jsonl
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
from deepsense6g_tii_amd.synthetic import make_batch
from oracle import fusion_ref as fr
from oracle import train_ref as tr

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--lr", type=float, default=1e-4)
ap.add_argument("--pool", type=int, default=16)
ap.add_argument("--eval-batches", type=int, default=4)
ap.add_argument("--eval-every", type=int, default=25)
ap.add_argument("--threads", type=int, default=0)
args = ap.parse_args()
if args.threads:
    torch.set_num_threads(args.threads)

torch.manual_seed(100)
cfg = GlobalConfig()
init = TransFuser(cfg, "cpu").state_dict()          # parameter containers only: the very init the HIP run starts from
rcfg = fr.RefConfig()
sd = {k: (v.clone().float().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
      for k, v in init.items()}
params = [v for v in sd.values() if v.requires_grad]
opt = torch.optim.AdamW(params, lr=args.lr)          # train2_seq.py:539 defaults (wd 0.01)


def evaluate():
    preds, truth = [], []
    with torch.no_grad():
        for i in range(args.eval_batches):
            f, l, r, g, t, beam = make_batch(args.batch, seed=10_000 + i, learnable=True)
            lg = fr.transfuser_forward(sd, f, l, r, g, rcfg, fr.Ctx(training=False))
            preds.append(torch.argsort(lg, dim=1, descending=True).numpy())
            truth.append(beam.numpy())
    p, y = np.concatenate(preds), np.concatenate(truth)
    return tr.compute_dba_score(p, y), tr.compute_acc(p, y)


dba, acc = evaluate()
print(json.dumps(dict(step=0, dba=dba, top123=acc.tolist(), path="cpu-oracle")), flush=True)
t0 = time.time()
for step in range(1, args.steps + 1):
    f, l, r, g, t, _ = make_batch(args.batch, seed=100 + step % args.pool, learnable=True)
    opt.zero_grad(set_to_none=True)
    logits = fr.transfuser_forward(sd, f, l, r, g, rcfg, fr.Ctx(training=True, dropout=True))
    loss = tr.sigmoid_focal_loss(logits, t)
    loss.backward()
    opt.step()
    if step % args.eval_every == 0 or step == args.steps:
        dba, acc = evaluate()
        print(json.dumps(dict(step=step, loss=float(loss), dba=dba, top123=acc.tolist(), path="cpu-oracle",
                              samples_per_s=step * args.batch / (time.time() - t0))), flush=True)
