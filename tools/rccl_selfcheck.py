"""One-rank RCCL self-check (a one-GPU box cannot host two RCCL ranks): initialises the "nccl" backend with
world_size 1 and drives the bucketed gradient all-reduce path of dist.GradReducer through a training iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from deepsense6g_tii_amd import dist as ddist
from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
from deepsense6g_tii_amd.synthetic import make_batch
from deepsense6g_tii_amd.train import FusedAdamW, train_iteration

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda:0")
model = TransFuser(GlobalConfig(n_layer=2), dev); model.train()
opt = FusedAdamW(model, lr=1e-4)
ddist.broadcast_parameters(model)
red = ddist.attach(model, opt, min_bucket_elems=1 << 20)
red_world = red.world
batch = make_batch(2, seed=1, device=dev)[:5]
# force the collective path even at world 1
import deepsense6g_tii_amd.dist as D
orig = D.GradReducer._flush
def flush(self):
    if self.hi > self.lo:
        self.works.append(dist.all_reduce(self.g[self.lo:self.hi], op=dist.ReduceOp.SUM, async_op=True))
        self.issued.append((self.lo, self.hi)); self.lo = self.hi
D.GradReducer._flush = flush
g_ref = None
for it in range(3):
    loss, _ = train_iteration(model, opt, batch, None, red)
torch.cuda.synchronize()
print("rccl world", red_world, "buckets", len(red.issued), "loss", float(loss), "OK")
dist.destroy_process_group()
