"""Inference (validation / serving) forward of the eval-mode model: BN folded into the convs vs separate BN kernels at
bs=12, and single-sample latency eager vs replayed from a captured HIP graph (TransFuser.capture_inference)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
from deepsense6g_tii_amd.synthetic import make_batch

dev = torch.device("cuda:0")
model = TransFuser(GlobalConfig(), dev).eval()


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for B in (12, 1):
    batch = make_batch(B, seed=100, device=dev)[:4]
    with torch.no_grad():
        for fold in (True, False):
            model.fold_bn_eval = fold
            dt = timeit(lambda: model(*batch))
            print(f"B={B:2d} fold_bn_eval={fold}: {dt * 1e3:6.2f} ms / forward, {B / dt:6.0f} samples/s", flush=True)
        model.fold_bn_eval = True
        if hasattr(model, "capture_inference"):
            runner = model.capture_inference(*batch)
            ref = model(*batch)
            out = runner(*batch)
            assert torch.equal(out, ref), (out - ref).abs().max()
            dt = timeit(lambda: runner(*batch))
            print(f"B={B:2d} HIP graph replay     : {dt * 1e3:6.2f} ms / forward, {B / dt:6.0f} samples/s", flush=True)
