"""Per-phase shader clocks of attn_bwd_dkv_kernel (wave 0 of workgroup 0), from the -DDS6G_ATTN_CLOCKS debug build:
    make -C deepsense6g_tii_amd/csrc ../libds6g_attnclk.so && DS6G_LIB=deepsense6g_tii_amd/libds6g_attnclk.so python tools/attn_clocks.py
Prints, per head dim, cycles per tile step in each phase (B = 12, T = 962, 4 heads, dropout P env, default 0.1)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd import ops
from deepsense6g_tii_amd._lib import lib, LIB_PATH

dev = torch.device("cuda:0")
raw = ctypes.CDLL(os.environ.get("DS6G_LIB", LIB_PATH))
raw.ds6g_attn_clocks_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
ws = ops.Workspace(dev, 2 << 30)
B, T, nh = 12, 962, 4
p = float(os.environ.get("P", "0.1"))
names = ["dma issue", "S = Q K^T", "dP = dO V^T", "elementwise", "hand-over stores", "dV", "dK", "wait DMA", "barrier"]
for hd in (16, 32, 64, 128):
    C = nh * hd
    q, k, v, do = (torch.randn(B * T, C, device=dev) for _ in range(4))
    o, lse = ops.attention_fwd(q, k, v, B, T, nh, ws, p, 1, 0)
    ops.attention_bwd(q, k, v, o, do, lse, B, T, nh, ws, p, 1, 0)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert raw.ds6g_attn_clocks_read(buf, 1) == 0
    ops.attention_bwd(q, k, v, o, do, lse, B, T, nh, ws, p, 1, 0)
    torch.cuda.synchronize()
    assert raw.ds6g_attn_clocks_read(buf, 1) == 0
    steps = max(1, buf[15])
    tot = sum(buf[i] for i in range(9))
    print(f"hd={hd}: {steps} steps counted (all dkv launches of one backward), {tot / steps:.0f} clocks per step")
    for i, n in enumerate(names):
        print(f"    {n:18s} {buf[i] / steps:9.0f}  {100.0 * buf[i] / max(1, tot):5.1f} %")
