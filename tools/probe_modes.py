"""Per-kernel error of the three matrix-core modes against fp32 torch (CPU) on workload conv shapes (GPU box)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from deepsense6g_tii_amd import ops

dev = torch.device("cuda:0")
ws = ops.Workspace(dev, 256 << 20)


def relerr(a, b):
    return float((a.cpu() - b).abs().max() / b.abs().max())


shapes = [(10, 16, 256, 512, 3, 2, 1), (10, 8, 512, 512, 3, 1, 1), (10, 16, 256, 512, 1, 2, 0), (10, 64, 64, 64, 3, 1, 1),
          (10, 32, 128, 128, 3, 1, 1), (10, 64, 64, 128, 3, 2, 1)]
for (N, H, C, K, R, st, pd) in shapes:
    g = torch.Generator().manual_seed(H + C)
    x = torch.randn(N, C, H, H, generator=g, requires_grad=True)
    w = (torch.randn(K, C, R, R, generator=g) / math.sqrt(C * R * R)).requires_grad_(True)
    y = F.conv2d(x, w, None, st, pd)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    wg = w.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    for mode in ("f32", "f32x6", "f32x3", "bf16"):
        ops.set_compute_mode(mode)
        yg = ops.conv2d_fwd(xg, wg.data_ptr(), K, R, R, st, pd)
        dx = ops.conv2d_dgrad(dyg, wg.data_ptr(), tuple(xg.shape), R, R, st, pd)
        dw = torch.empty_like(wg)
        ops.conv2d_wgrad(xg, dyg, dw.data_ptr(), R, R, st, pd, ws)
        print(f"N{N} H{H} C{C} K{K} R{R} s{st} {mode:6s} fwd {relerr(yg.permute(0, 3, 1, 2), y.detach()):.2e} "
              f"dgrad {relerr(dx.permute(0, 3, 1, 2), x.grad):.2e} wgrad {relerr(dw.permute(0, 3, 1, 2), w.grad):.2e}", flush=True)
ops.set_compute_mode("f32")
