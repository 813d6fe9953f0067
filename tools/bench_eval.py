"""Inference (validation) throughput of the eval-mode forward, bs=12: BN folded into the convs vs separate BN kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
from deepsense6g_tii_amd.synthetic import make_batch

dev = torch.device("cuda:0")
model = TransFuser(GlobalConfig(), dev).eval()
batch = make_batch(12, seed=100, device=dev)[:4]
for fold in (True, False):
    model.fold_bn_eval = fold
    with torch.no_grad():
        for _ in range(3): model(*batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): model(*batch)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"fold_bn_eval={fold}: {dt * 1e3:.1f} ms / forward, {12 / dt:.0f} samples/s")
