"""Micro-benchmark of bgemm.hip (bf16-stored operands) beside the fp32-storage kernel in bf16 matrix-core mode (mode 1) on
the layer shapes of the bs=12 workload (GPU box).  Prints us and TFLOP/s per product for both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd import ops
from deepsense6g_tii_amd._lib import lib

dev = torch.device("cuda:0")
ws = ops.Workspace(dev, 512 << 20)
N = 60
SHAPES = [("l1_3x3", N, 64, 64, 64, 64, 3, 1, 1), ("l2_3x3s2", N, 64, 64, 64, 128, 3, 2, 1), ("l2_3x3", N, 32, 32, 128, 128, 3, 1, 1),
          ("l3_3x3", N, 16, 16, 256, 256, 3, 1, 1), ("l4_3x3", N, 8, 8, 512, 512, 3, 1, 1), ("l4_1x1s2", N, 16, 16, 256, 512, 1, 2, 0)]
LIN = [("gpt1_qkv", 11544, 192, 64), ("gpt1_fc1", 11544, 256, 64), ("gpt1_fc2", 11544, 64, 256), ("gpt2_fc1", 11544, 512, 128),
       ("gpt3_fc1", 11544, 1024, 256), ("gpt4_qkv", 11544, 1536, 512), ("gpt4_proj", 11544, 512, 512),
       ("gpt4_fc1", 11544, 2048, 512), ("gpt4_fc2", 11544, 512, 2048), ("sq4096", 4096, 4096, 4096)]
reps = int(os.environ.get("REPS", "10"))
only = os.environ.get("ONLY")
BF = torch.bfloat16


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'shape':12s} {'mode':6s} {'bf16-stored us':>14s} {'TF/s':>7s} | {'fp32-stored mode1 us':>20s} {'TF/s':>7s}")
for name, n, H, W, C, K, R, st, pad in SHAPES:
    if only and only not in name: continue
    x = torch.randn(n, H, W, C, device=dev); w = torch.randn(K, R, R, C, device=dev) * 0.05
    Ho, Wo = ops.conv_out_hw(H, W, R, R, st, pad)
    dy = torch.randn(n, Ho, Wo, K, device=dev)
    x16, w16, dy16 = x.to(BF), w.to(BF), dy.to(BF)
    dw = torch.empty_like(w)
    fl = 2.0 * n * Ho * Wo * K * R * R * C
    rows = [("fwd", lambda: ops.bf16_conv2d_fwd(x16, w16.data_ptr(), K, R, R, st, pad),
             lambda: ops.conv2d_fwd(x, w.data_ptr(), K, R, R, st, pad)),
            ("wgrad", lambda: ops.bf16_conv2d_wgrad(x16, dy16, dw.data_ptr(), R, R, st, pad, ws),
             lambda: ops.conv2d_wgrad(x, dy, dw.data_ptr(), R, R, st, pad, ws))]
    if st == 1:
        rows.insert(1, ("dgrad", lambda: ops.bf16_conv2d_dgrad(dy16, w16.data_ptr(), tuple(x.shape), R, R, st, pad),
                        lambda: ops.conv2d_dgrad(dy, w.data_ptr(), tuple(x.shape), R, R, st, pad)))
    for mode, f16, f32 in rows:
        lib().set_compute_mode(0)
        a = timeit(f16)
        lib().set_compute_mode(1)
        b = timeit(f32)
        lib().set_compute_mode(0)
        print(f"{name:12s} {mode:6s} {a:14.1f} {fl / a / 1e6:7.1f} | {b:20.1f} {fl / b / 1e6:7.1f}", flush=True)
for name, M, Nn, K in LIN:
    if only and only not in name: continue
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) * 0.05; b_ = torch.randn(Nn, device=dev)
    dy = torch.randn(M, Nn, device=dev); dw = torch.empty_like(w)
    x16, w16, dy16 = x.to(BF), w.to(BF), dy.to(BF)
    fl = 2.0 * M * Nn * K
    for mode, f16, f32 in (("fwd", lambda: ops.bf16_linear_fwd(x16, w16.data_ptr(), b_.data_ptr(), Nn),
                            lambda: ops.linear_fwd(x, w.data_ptr(), b_.data_ptr(), Nn)),
                           ("dgrad", lambda: ops.bf16_linear_dgrad(dy16, w16.data_ptr(), K),
                            lambda: ops.linear_dgrad(dy, w.data_ptr(), K)),
                           ("wgrad", lambda: ops.bf16_linear_wgrad(x16, dy16, dw.data_ptr(), ws),
                            lambda: ops.linear_wgrad(x, dy, dw.data_ptr(), ws))):
        lib().set_compute_mode(0)
        a = timeit(f16)
        lib().set_compute_mode(1)
        b = timeit(f32)
        lib().set_compute_mode(0)
        print(f"{name:12s} {mode:6s} {a:14.1f} {fl / a / 1e6:7.1f} | {b:20.1f} {fl / b / 1e6:7.1f}", flush=True)
