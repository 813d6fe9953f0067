"""cProfile of the Python host enqueueing training steps (no sync inside): where do the ~2000 launches per step spend host
time?  DTYPE=f32|bf16, STEPS=5.  Prints enqueue ms/step and the top functions by own time."""
import cProfile, io, os, pstats, sys, time
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
n = int(os.environ.get("STEPS", "5"))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(n): train_iteration(model, opt, batch)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"enqueue under cProfile {1e3 * (t1 - t0) / n:.1f} ms/step")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
