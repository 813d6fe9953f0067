"""Micro-benchmark of the implicit-GEMM kernel on the layer shapes of the bs=12 workload (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd import ops
from deepsense6g_tii_amd._lib import lib

dev = torch.device("cuda:0")
ws = ops.Workspace(dev, 256 << 20)
N = 60
SHAPES = [  # name, N, H, W, C, K, R, stride, pad
    ("stem7x7", N, 256, 256, 4, 64, 7, 2, 3),
    ("l1_3x3", N, 64, 64, 64, 64, 3, 1, 1),
    ("l2_3x3s2", N, 64, 64, 64, 128, 3, 2, 1),
    ("l2_3x3", N, 32, 32, 128, 128, 3, 1, 1),
    ("l3_3x3", N, 16, 16, 256, 256, 3, 1, 1),
    ("l4_3x3", N, 8, 8, 512, 512, 3, 1, 1),
    ("l4_1x1s2", N, 16, 16, 256, 512, 1, 2, 0),
]
LIN = [("bal768", 12288, 1024, 512), ("bal768k4096", 12288, 1024, 4096), ("bal1536", 12288, 2048, 512), ("gpt1_qkv", 11544, 64, 64), ("gpt1_fc1", 11544, 256, 64), ("gpt2_fc1", 11544, 512, 128),
       ("gpt3_fc1", 11544, 1024, 256), ("gpt4_qkv", 11544, 512, 512), ("gpt4_fc1", 11544, 2048, 512),
       ("gpt4_fc2", 11544, 512, 2048)]
reps = int(os.environ.get("REPS", "5"))
only = os.environ.get("ONLY")
if os.environ.get("ZERO"):  # all-zero operands: same instruction stream, minimal data toggling (power experiment)
    torch.randn = lambda *a, **k: torch.zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "generator"})
lib().set_debug_flags(int(os.environ.get("DBG", "0"), 0))
lib().set_compute_mode(int(os.environ.get("BF16", "0")))

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us

print(f"{'shape':12s} {'mode':6s} {'us':>9s} {'TFLOP/s':>8s} variant")
for name, n, H, W, C, K, R, st, pad in SHAPES:
    if only and only not in name: continue
    x = torch.randn(n, H, W, C, device=dev)
    w = torch.randn(K, R, R, C, device=dev) * 0.05
    Ho, Wo = ops.conv_out_hw(H, W, R, R, st, pad)
    y = torch.empty(n, Ho, Wo, K, device=dev)
    dy = torch.randn(n, Ho, Wo, K, device=dev)
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * n * Ho * Wo * K * R * R * C
    for mode, fn in (("fwd", lambda: ops.conv2d_fwd(x, w.data_ptr(), K, R, R, st, pad, out=y)),
                     ("dgrad", lambda: ops.conv2d_dgrad(dy, w.data_ptr(), tuple(x.shape), R, R, st, pad, out=dx)),
                     ("wgrad", lambda: ops.conv2d_wgrad(x, dy, dw.data_ptr(), R, R, st, pad, ws))):
        us = timeit(fn)
        print(f"{name:12s} {mode:6s} {us:9.1f} {fl / us / 1e6:8.1f} {lib().last_igemm_variant()}", flush=True)
for name, M, Nn, K in LIN:
    if only and only not in name: continue
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) * 0.05
    b = torch.randn(Nn, device=dev); y = torch.empty(M, Nn, device=dev)
    dy = torch.randn(M, Nn, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * M * Nn * K
    for mode, fn in (("fwd", lambda: ops.linear_fwd(x, w.data_ptr(), b.data_ptr(), Nn, out=y)),
                     ("dgrad", lambda: ops.linear_dgrad(dy, w.data_ptr(), K, out=dx)),
                     ("wgrad", lambda: ops.linear_wgrad(x, dy, dw.data_ptr(), ws))):
        us = timeit(fn)
        print(f"{name:12s} {mode:6s} {us:9.1f} {fl / us / 1e6:8.1f} {lib().last_igemm_variant()}", flush=True)
