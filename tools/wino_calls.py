"""Per-shape times of the Winograd forward / data-gradient launches inside one bs=12 training step (single stream, HIP
events around each call), beside the same shapes timed back to back in tools/bench_winograd.py."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd import ops
from deepsense6g_tii_amd._lib import lib
from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
from deepsense6g_tii_amd.synthetic import make_batch
from deepsense6g_tii_amd.train import FusedAdamW, train_iteration

dev = torch.device("cuda:0")
model = TransFuser(GlobalConfig(), dev); model.train(); model.multi_stream = os.environ.get("MULTI", "0") == "1"
opt = FusedAdamW(model, lr=1e-4)
batch = make_batch(12, seed=100, device=dev)[:5]
for _ in range(2): train_iteration(model, opt, batch)
torch.cuda.synchronize()
L = lib(); recs = []
orig = L.conv3x3_winograd_fwd
def call(*a):
    st = torch.cuda.ExternalStream(a[-1]) if a[-1] else torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st); orig(*a); e1.record(st)
    recs.append((tuple(a[3:9]), e0, e1))
L.conv3x3_winograd_fwd = call
for _ in range(3): train_iteration(model, opt, batch)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1 in recs:
    d = agg.setdefault(key, [0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1) * 1e3
print("N, H, W, C, K, accumulate: launches per step, average us")
for key, (n, us) in sorted(agg.items()):
    print(key, n // 3, f"{us / n:7.1f}")
