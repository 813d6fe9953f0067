"""Where the multi-stream step spends its wall time, phase by phase: HIP events on the calling stream at the entry / exit of
every GPT stage's forward and backward (the trunk layers run between them on the three trunk streams and are joined before
a stage starts), and at the step's start / after the optimizer.  DTYPE=f32|bf16.  Prints ms per phase, averaged over STEPS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd import ops
from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
from deepsense6g_tii_amd.synthetic import make_batch
from deepsense6g_tii_amd.train import FusedAdamW, train_iteration

dev = torch.device("cuda:0")
ops.set_compute_mode(os.environ.get("DTYPE", "f32"))
model = TransFuser(GlobalConfig(), dev); model.train()
opt = FusedAdamW(model, lr=1e-4)
batch = make_batch(12, seed=100, device=dev)[:5]
for _ in range(3): train_iteration(model, opt, batch)
torch.cuda.synchronize()
marks = []
def ev(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e))
of, ob = model._stage_fwd, model._stage_bwd
cnt = {"f": 0, "b": 0}
def sf(*a, **k):
    cnt["f"] += 1; s = cnt["f"]
    ev(f"trunk fwd -> GPT{s}"); r = of(*a, **k); ev(f"GPT{s} fwd"); return r
def sb(*a, **k):
    s = 4 - cnt["b"]; cnt["b"] += 1
    ev(f"trunk bwd -> GPT{s}"); r = ob(*a, **k); ev(f"GPT{s} bwd"); return r
model._stage_fwd, model._stage_bwd = sf, sb
n = int(os.environ.get("STEPS", "5"))
acc = {}
order = []
for it in range(n):
    marks.clear(); cnt["f"] = cnt["b"] = 0
    ev("start"); train_iteration(model, opt, batch); ev("end (stems bwd + AdamW)")
    torch.cuda.synchronize()
    for (t0, e0), (t1, e1) in zip(marks[:-1], marks[1:]):
        key = t1 if t1 not in ("end (stems bwd + AdamW)",) else t1
        key = f"{len([k for k in order if k.startswith(t1)]) if it == 0 and False else ''}{t1}"
        acc.setdefault((marks.index((t1, e1)), t1), []).append(e0.elapsed_time(e1))
tot = 0.0
for (i, tag), v in sorted(acc.items()):
    ms = sum(v) / len(v); tot += ms
    print(f"{i:3d} {tag:28s} {ms:7.2f} ms")
print(f"    total {tot:.2f} ms")
