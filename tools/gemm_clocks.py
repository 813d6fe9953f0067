"""Per-phase shader clocks and per-workgroup entry / exit stamps of igemm_kernel (fp32) and bgemm_kernel (bf16 storage),
from the -DDS6G_GEMM_CLOCKS debug build:
    make -C deepsense6g_tii_amd/csrc ../libds6g_gemmclk.so && DS6G_LIB=deepsense6g_tii_amd/libds6g_gemmclk.so python tools/gemm_clocks.py
For every case: cycles of ONE wave (wave 0 of the middle workgroup) per phase - set-up, first-tile latency, then per k-tile:
DMA issue / fragment reads + MFMA chain / DMA wait / barrier - and epilogue; and over all workgroups: when they entered the
kernel (dispatch ramp), how long each lived, when the last one left."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepsense6g_tii_amd import ops
from deepsense6g_tii_amd._lib import LIB_PATH

dev = torch.device("cuda:0")
raw = ctypes.CDLL(os.environ.get("DS6G_LIB", LIB_PATH))
NWG = 16384
for fn in (raw.ds6g_igemm_clocks_read, raw.ds6g_bgemm_clocks_read):
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
ws = ops.Workspace(dev, 1 << 30)
BF = torch.bfloat16
M = 11544
names = ["set-up", "first tile(s)", "DMA issue", "frag reads + MFMA", "DMA wait", "barrier", "epilogue"]


def report(tag, read, fn, mfma_cycles_per_ktile):
    fn(); fn(); torch.cuda.synchronize()
    clk = (ctypes.c_ulonglong * 16)(); wg = (ctypes.c_ulonglong * (2 * NWG))()
    assert read(clk, wg, NWG, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    assert read(clk, wg, NWG, 1) == 0
    w = np.array(wg[:], dtype=np.int64).reshape(NWG, 2)
    w = w[(w[:, 0] > 0) & (w[:, 1] > 0)]
    nk = max(1, clk[15])
    t0 = w[:, 0].min()
    st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0     # us (100 MHz real-time counter)
    life = en - st
    tot = clk[14]
    print(f"{tag}: event time {e0.elapsed_time(e1) * 1e3:.1f} us (incl. any split-K reduction); {len(w)} workgroups; "
          f"instrumented wave: {tot} cycles, {nk} k-tiles")
    print(f"    workgroup entry: p50 {np.percentile(st, 50):5.1f}  p90 {np.percentile(st, 90):5.1f}  max {st.max():5.1f} us | "
          f"life: mean {life.mean():5.1f}  p10 {np.percentile(life, 10):5.1f}  p90 {np.percentile(life, 90):5.1f} us | last exit {en.max():5.1f} us")
    loop = sum(clk[i] for i in (2, 3, 4, 5))
    print(f"    set-up {clk[0]}  first tile(s) {clk[1]}  epilogue {clk[6]}  loop {loop} = {loop / nk:.0f} per k-tile "
          f"(MFMA alone {mfma_cycles_per_ktile}): issue {clk[2] / nk:.0f}  reads+MFMA {clk[3] / nk:.0f}  wait {clk[4] / nk:.0f}  barrier {clk[5] / nk:.0f}")


def lin32(N, K):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev); dw = torch.empty_like(w)
    return (lambda: ops.linear_fwd(x, w.data_ptr(), b.data_ptr(), N), lambda: ops.linear_dgrad(dy, w.data_ptr(), K),
            lambda: ops.linear_wgrad(x, dy, dw.data_ptr(), ws))


def lin16(N, K):
    x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF); b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev).to(BF); dw = torch.empty(N, K, device=dev)
    return (lambda: ops.bf16_linear_fwd(x, w.data_ptr(), b.data_ptr(), N), lambda: ops.bf16_linear_dgrad(dy, w.data_ptr(), K),
            lambda: ops.bf16_linear_wgrad(x, dy, dw.data_ptr(), ws))


which = os.environ.get("WHICH", "f32,bf16").split(",")
for N, K in ((2048, 512), (512, 512), (512, 128), (64, 64)):
    if "f32" in which:
        for mode, fn in zip(("fwd", "dgrad", "wgrad"), lin32(N, K)):
            # 64x64 tile: 8 MFMAs 32x32x2 (64 cycles) per 16-deep k-tile; 128x64 wgrad: 16
            report(f"fp32 linear {mode} M={M} N={N} K={K}", raw.ds6g_igemm_clocks_read, fn, "512-1024")
    if "bf16" in which:
        for mode, fn in zip(("fwd", "dgrad", "wgrad"), lin16(N, K)):
            report(f"bf16 linear {mode} M={M} N={N} K={K}", raw.ds6g_bgemm_clocks_read, fn, "512 (128x128) / 128 (64x64)")
if "conv" in which:
    for (n, H, W, C, Kc, R, st, pad) in ((60, 64, 64, 64, 64, 3, 1, 1), (60, 32, 32, 128, 128, 3, 1, 1), (60, 16, 16, 256, 256, 3, 1, 1)):
        x16 = torch.randn(n, H, W, C, device=dev).to(BF); w16 = (torch.randn(Kc, R, R, C, device=dev) * 0.05).to(BF)
        dy16 = torch.randn(n, H, W, Kc, device=dev).to(BF); dw = torch.empty(Kc, R, R, C, device=dev)
        report(f"bf16 conv fwd {n}x{H}x{W}x{C}->{Kc}", raw.ds6g_bgemm_clocks_read, lambda: ops.bf16_conv2d_fwd(x16, w16.data_ptr(), Kc, R, R, st, pad), "512 / 128")
        report(f"bf16 conv dgrad", raw.ds6g_bgemm_clocks_read, lambda: ops.bf16_conv2d_dgrad(dy16, w16.data_ptr(), tuple(x16.shape), R, R, st, pad), "512 / 128")
        report(f"bf16 conv wgrad", raw.ds6g_bgemm_clocks_read, lambda: ops.bf16_conv2d_wgrad(x16, dy16, dw.data_ptr(), R, R, st, pad, ws), "512 / 128")
