"""winograd_pc_kernel per layer shape (N = 60): time of the forward and of the accumulating data-gradient form, for the
item order chosen by DS6G_PC_XCD (unset = auto, 0 = plain order, 1 / 2 / 4 / 8 = forced channel-group count).  Under
`rocprofv3 --pmc FETCH_SIZE --kernel-trace` the per-dispatch counters give the L2-miss read traffic per shape
(tools/pc_xcd_traffic.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd._lib import lib

dev = torch.device("cuda:0")
L = lib()
st = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get("REPS", "20"))
print("DS6G_PC_XCD =", os.environ.get("DS6G_PC_XCD", "(auto)"))
for name, N, H, C in (("l1 64x64 c64", 60, 64, 64), ("l2 32x32 c128", 60, 32, 128), ("l3 16x16 c256", 60, 16, 256), ("l4 8x8 c512", 60, 8, 512)):
    K = C
    x = torch.randn(N, H, H, C, device=dev)
    w = torch.randn(K, 3, 3, C, device=dev) * (1.0 / (3 * C ** 0.5))
    u = torch.empty(L.winograd_weight_floats(K, C), device=dev)
    y = torch.empty(N, H, H, K, device=dev)
    L.winograd_weights(w.data_ptr(), u.data_ptr(), K, C, 0, st)
    out = []
    for acc in (0, 1):
        fn = lambda: L.conv3x3_winograd_fwd(x.data_ptr(), u.data_ptr(), y.data_ptr(), N, H, H, C, K, acc, st)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    print(f"{name:16s} fwd {out[0]:7.1f} us   accumulate {out[1]:7.1f} us   x {x.numel()*4/1e6:6.1f} MB  U {u.numel()*4/1e6:5.1f} MB", flush=True)
