"""Does running the three (independent) trunks on three HIP streams fill launch tails?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd import ops
dev = torch.device("cuda:0")
N = 60
def mk(H, C, K):
    x = torch.randn(N, H, H, C, device=dev); w = torch.randn(K, 3, 3, C, device=dev) * 0.05
    y = torch.empty(N, H, H, K, device=dev)
    mean = torch.zeros(K, device=dev); inv = torch.ones(K, device=dev); g = torch.ones(K, device=dev); b = torch.zeros(K, device=dev)
    y2 = torch.empty_like(y)
    return x, w, y, mean, inv, g, b, y2
for name, H, C in (("layer1", 64, 64), ("layer2", 32, 128), ("layer3", 16, 256), ("layer4", 8, 512)):
    sets = [mk(H, C, C) for _ in range(3)]
    wss = [ops.Workspace(dev, 64 << 20) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    def chain(s, ws):  # conv -> bn stats -> bn apply, x4 (a BasicBlock-ish sequence of dependent launches)
        x, w, y, mean, inv, g, b, y2 = s
        for _ in range(4):
            ops.conv2d_fwd(x, w.data_ptr(), C, 3, 3, 1, 1, out=y)
            ops.bn_stats(N * H * H, C, y, mean, inv, 0, 0, ws)
            ops.bn_apply(y, mean, inv, g.data_ptr(), b.data_ptr(), True, out=y2)
    def serial():
        for s, ws in zip(sets, wss): chain(s, ws)
    def parallel():
        cur = torch.cuda.current_stream()
        for st in streams: st.wait_stream(cur)
        for s, ws, st in zip(sets, wss, streams):
            with torch.cuda.stream(st): chain(s, ws)
        for st in streams: cur.wait_stream(st)
    for fn, label in ((serial, "1 stream "), (parallel, "3 streams")):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name} {label}: {e0.elapsed_time(e1) / 5 * 1e3:9.1f} us", flush=True)
