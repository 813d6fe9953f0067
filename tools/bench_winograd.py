"""Winograd F(2x2,3x3) conv vs the direct implicit GEMM on the 3x3 / stride-1 layer shapes (correctness + time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from deepsense6g_tii_amd import ops
from deepsense6g_tii_amd._lib import lib

dev = torch.device("cuda:0")
L = lib()
L.set_debug_flags(int(os.environ.get('DBG', '0')))
st = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get("REPS", "20"))
ACC = 0  # timing runs: y += conv (the data-gradient form) when ACC=1 in the environment


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'shape':22s} {'direct us':>10s} {'wino us':>9s} {'wino+U us':>10s} {'max rel err':>12s}")
for name, N, H, C, K in (("l1 64x64 c64", 60, 64, 64, 64), ("l2 32x32 c128", 60, 32, 128, 128), ("l3 16x16 c256", 60, 16, 256, 256),
                         ("l4 8x8 c512", 60, 8, 512, 512), ("small 8x8", 3, 8, 32, 32)):
    x = torch.randn(N, H, H, C, device=dev)
    w = torch.randn(K, 3, 3, C, device=dev) * (1.0 / (3 * C ** 0.5))
    assert L.winograd_supported(N, H, H, C, K), name
    u = torch.empty(L.winograd_weight_floats(K, C), device=dev)
    y = torch.empty(N, H, H, K, device=dev)
    def wino(with_u=False):
        global ACC
        if with_u: L.winograd_weights(w.data_ptr(), u.data_ptr(), K, C, 0, st)
        L.conv3x3_winograd_fwd(x.data_ptr(), u.data_ptr(), y.data_ptr(), N, H, H, C, K, ACC, st)
    L.winograd_weights(w.data_ptr(), u.data_ptr(), K, C, 0, st)
    wino()
    ref = ops.conv2d_fwd(x, w.data_ptr(), K, 3, 3, 1, 1)
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    if N * H * H * C < 3e6:  # also against torch on the CPU for the small case
        rt = F.conv2d(x.cpu().permute(0, 3, 1, 2), w.cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
        err = max(err, ((y.cpu() - rt).abs().max() / rt.abs().max()).item())
    # dgrad through the transposed / flipped filter
    dy = torch.randn(N, H, H, K, device=dev)
    ud = torch.empty(L.winograd_weight_floats(C, K), device=dev)
    dx = torch.empty(N, H, H, C, device=dev)
    L.winograd_weights(w.data_ptr(), ud.data_ptr(), K, C, 1, st)
    L.conv3x3_winograd_fwd(dy.data_ptr(), ud.data_ptr(), dx.data_ptr(), N, H, H, K, C, 0, st)
    dref = ops.conv2d_dgrad(dy, w.data_ptr(), (N, H, H, C), 3, 3, 1, 1)
    derr = ((dx - dref).abs().max() / dref.abs().max()).item()
    td = timeit(lambda: ops.conv2d_fwd(x, w.data_ptr(), K, 3, 3, 1, 1, out=ref))
    ACC = int(os.environ.get("ACC", "0"))
    tw = timeit(lambda: wino(False))
    twu = timeit(lambda: wino(True))
    print(f"{name:22s} {td:10.1f} {tw:9.1f} {twu:10.1f} {err:12.2e}  dgrad err {derr:.2e}", flush=True)

print("\nweight gradient: direct vs Winograd")
ws = ops.Workspace(dev, 1 << 30)
for name, N, H, C, K in (("l1 64x64 c64", 60, 64, 64, 64), ("l2 32x32 c128", 60, 32, 128, 128), ("l3 16x16 c256", 60, 16, 256, 256),
                         ("l4 8x8 c512", 60, 8, 512, 512)):
    x = torch.randn(N, H, H, C, device=dev)
    dy = torch.randn(N, H, H, K, device=dev)
    d1 = torch.empty(K, 3, 3, C, device=dev)
    d2 = torch.empty(K, 3, 3, C, device=dev)
    ops.conv2d_wgrad(x, dy, d1.data_ptr(), 3, 3, 1, 1, ws)
    ops.conv3x3_winograd_wgrad(x, dy, d2.data_ptr(), ws)
    err = ((d1 - d2).abs().max() / d1.abs().max()).item()
    t1 = timeit(lambda: ops.conv2d_wgrad(x, dy, d1.data_ptr(), 3, 3, 1, 1, ws))
    t2 = timeit(lambda: ops.conv3x3_winograd_wgrad(x, dy, d2.data_ptr(), ws))
    print(f"{name:22s} {t1:10.1f} {t2:9.1f}   max rel err {err:.2e}", flush=True)
