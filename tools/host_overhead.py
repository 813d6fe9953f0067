"""How far ahead of the GPU does the Python host run?  Enqueue time per step (no sync) vs completed time per step."""
import os, sys, time
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
n = 10
t0 = time.perf_counter()
for _ in range(n): train_iteration(model, opt, batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / n:.1f} ms/step, complete {1e3 * (t2 - t0) / n:.1f} ms/step, threads={torch.get_num_threads()}")
# the enqueue wall time above includes back-pressure (hipLaunchKernel blocks when the launch queues are full, i.e. when the
# host is FAR ahead).  What the host itself needs per step: enqueue ONE step into an idle GPU, several times
cost = []
for _ in range(6):
    torch.cuda.synchronize()
    a = time.perf_counter()
    train_iteration(model, opt, batch)
    cost.append(time.perf_counter() - a)
    torch.cuda.synchronize()
cost.sort()
print(f"host cost of enqueueing one step into an idle GPU: median {1e3 * cost[len(cost) // 2]:.1f} ms, min {1e3 * cost[0]:.1f} ms")
