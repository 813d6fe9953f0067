"""Micro-benchmark of the fused attention kernels at the bs=12 workload shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd import ops

dev = torch.device("cuda:0")
ws = ops.Workspace(dev, 1 << 30)
from deepsense6g_tii_amd._lib import lib
lib().set_compute_mode(int(os.environ.get('BF16', '0')))
lib().set_debug_flags(int(os.environ.get('DBG', '0'), 0))
B, T, nh = int(os.environ.get("B", "12")), 962, 4
reps = int(os.environ.get("REPS", "10"))

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

print(f"B={B}  {'hd':>4s} {'fwd us':>8s} {'TF':>6s} {'bwd us':>8s} {'TF(5 prod)':>10s}")
for hd in (16, 32, 64, 128):
    C = nh * hd
    q, k, v, do = (torch.randn(B * T, C, device=dev) for _ in range(4))
    p = float(os.environ.get("P", "0.1"))
    o, lse = ops.attention_fwd(q, k, v, B, T, nh, ws, p, 1, 0)
    f = timeit(lambda: ops.attention_fwd(q, k, v, B, T, nh, ws, p, 1, 0))
    b = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, B, T, nh, ws, p, 1, 0))
    prod = 2.0 * B * nh * T * T * hd  # flops of one T x T x hd product
    print(f"      {hd:4d} {f:8.1f} {2 * prod / f / 1e6:6.1f} {b:8.1f} {5 * prod / b / 1e6:10.1f}", flush=True)
