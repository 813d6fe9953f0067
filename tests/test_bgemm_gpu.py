"""bgemm.hip (conv / linear products on bf16-STORED operands, the real bf16 path) through the C ABI against torch on the CPU.
Inputs are drawn in fp32 and rounded to bf16 once; the reference multiplies exactly those bf16 values in fp64, so what is
compared is the kernel's arithmetic: exact bf16 x bf16 products, fp32 accumulation (bar 2e-5 of the largest output for an
fp32 result: accumulation order only), plus ONE rounding of the result to bf16 where the output is bf16 (bar 2^-8 relative
per element: half a bf16 ulp is 2^-9)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def r16(t):
    return t.to(BF)


def close32(a, b, tol=2e-5):
    a, b = a.double().cpu(), b.double().cpu()
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= tol * ref + 1e-30, f"max err {err:.3e} vs ref max {ref:.3e}"


def close16(a, b, roundings=1):
    """bf16 result against the exact value: every element within one bf16 rounding (+ accumulation noise); the
    accumulating forms round the product and the sum (two roundings, of values up to |product| + |old|)"""
    a, b = a.double().cpu(), b.double().cpu()
    bound = b.abs() * 2.0 ** -8 * roundings + 3e-5 * b.abs().max()
    bad = ((a - b).abs() > bound).sum().item()
    assert bad == 0, f"{bad} of {a.numel()} elements off by more than a bf16 rounding; worst {((a - b).abs() - bound).max().item():.3e}"


LIN = [(962 * 2, 64, 64), (12 * 962, 192, 64), (1000, 512, 2048), (777, 2048, 512), (300, 128, 256), (12 * 962, 1536, 512)]


@pytest.mark.parametrize("M,N,K", LIN)
def test_bf16_linear_fwd_dgrad_wgrad(dev, M, N, K):
    from deepsense6g_tii_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = r16(torch.randn(M, K, generator=g))
    w = r16(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    xg, wg, bg = x.cuda(), w.cuda(), b.cuda()
    xd, wd = x.double(), w.double()
    lin = xd @ wd.t() + b.double()
    ws = ops.Workspace(dev, 256 << 20)
    # forward: plain bf16 output; bias + ReLU -> bf16; bias + residual -> fp32
    close16(ops.bf16_linear_fwd(xg, wg.data_ptr(), 0, N), xd @ wd.t())
    close16(ops.bf16_linear_fwd(xg, wg.data_ptr(), bg.data_ptr(), N, relu=True), lin.relu())
    y32 = ops.bf16_linear_fwd(xg, wg.data_ptr(), bg.data_ptr(), N, residual=res.cuda())
    assert y32.dtype == torch.float32
    close32(y32, lin + res.double())
    # data gradient: bf16 dy, ReLU mask from a bf16 tensor, bf16 / fp32 outputs, accumulate
    dy = r16(torch.randn(M, N, generator=g))
    dyg = dy.cuda()
    dx_ref = dy.double() @ wd
    close16(ops.bf16_linear_dgrad(dyg, wg.data_ptr(), K), dx_ref)
    close32(ops.bf16_linear_dgrad(dyg, wg.data_ptr(), K, out16=False), dx_ref)
    msk = r16(torch.randn(M, K, generator=g))
    close16(ops.bf16_linear_dgrad(dyg, wg.data_ptr(), K, relu_mask_src=msk.cuda()), dx_ref * (msk.double() > 0))
    base = torch.randn(M, K, generator=g)
    acc = base.cuda().clone()
    ops.bf16_linear_dgrad(dyg, wg.data_ptr(), K, out16=False, out=acc, accumulate=True)
    close32(acc, dx_ref + base.double())
    # weight + bias gradient (fp32 outputs), and the accumulate form
    dw_ref, db_ref = dy.double().t() @ xd, dy.double().sum(0)
    dw = torch.full((N, K), float("nan"), device=dev)
    db = torch.full((N,), float("nan"), device=dev)
    ops.bf16_linear_wgrad(xg, dyg, dw.data_ptr(), ws, dbias_ptr=db.data_ptr())
    close32(dw, dw_ref, 3e-5)
    close32(db, db_ref, 3e-5)
    ops.bf16_linear_wgrad(xg, dyg, dw.data_ptr(), ws, accumulate=True, dbias_ptr=db.data_ptr())
    close32(dw, 2 * dw_ref, 3e-5)
    close32(db, 2 * db_ref, 3e-5)
    torch.cuda.synchronize()


def test_bf16_linear_dropout_epilogue_draws_the_fp32_paths_mask(dev):
    """the dropout counter of the epilogue is the element index, as in igemm.hip: same (seed, offset) -> same mask"""
    from deepsense6g_tii_amd import ops
    M, N, K = 2048, 256, 64
    g = torch.Generator().manual_seed(1)
    x = r16(torch.randn(M, K, generator=g)).cuda()
    w = r16(torch.randn(N, K, generator=g) / 8).cuda()
    res = torch.zeros(M, N, device=dev)
    y0 = ops.bf16_linear_fwd(x, w.data_ptr(), 0, N, residual=res)
    yd = ops.bf16_linear_fwd(x, w.data_ptr(), 0, N, residual=res, drop_p=0.25, seed=7, seed_off=4096)
    ref = ops.linear_fwd(x.float(), w.float().data_ptr(), 0, N, residual=res, drop_p=0.25, seed=7, seed_off=4096)
    keep = yd != 0
    assert abs(1 - keep.float().mean().item() - 0.25) < 0.01
    assert torch.equal(keep, ref != 0)
    assert torch.allclose(yd[keep], y0[keep] / 0.75, rtol=1e-6, atol=1e-6)


CONV = [
    # N, H, W, C, K, R, stride, pad
    (2, 16, 16, 64, 64, 3, 1, 1),
    (3, 16, 16, 64, 128, 3, 2, 1),      # strided forward / wgrad (its dgrad stays on the fp32-storage kernel)
    (2, 16, 16, 64, 128, 1, 2, 0),
    (5, 8, 8, 256, 512, 3, 1, 1),
    (1, 12, 32, 128, 192, 3, 1, 1),     # ragged M, N not multiples of the tiles
    (10, 64, 64, 64, 64, 3, 1, 1),      # 128 x 128 tiles
    (4, 32, 32, 128, 128, 3, 1, 1),
    (60, 8, 8, 512, 512, 3, 1, 1),      # benchmark batch, whole images per wgrad k-tile
]


@pytest.mark.parametrize("case", CONV, ids=lambda c: "x".join(map(str, c)))
def test_bf16_conv_fwd_dgrad_wgrad(dev, case):
    from deepsense6g_tii_amd import ops
    N, H, W, C, K, R, st, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = r16(torch.randn(N, C, H, W, generator=g)).double().requires_grad_(True)
    w = r16(torch.randn(K, C, R, R, generator=g) / math.sqrt(C * R * R)).double().requires_grad_(True)
    y = F.conv2d(x, w, None, st, pad)
    dy = r16(torch.randn(y.shape, generator=g)).double()
    y.backward(dy)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(BF).cuda()  # noqa: E731
    nchw = lambda t: t.cpu().permute(0, 3, 1, 2)  # noqa: E731
    xg, wg, dyg = nhwc(x), nhwc(w), nhwc(dy)
    ws = ops.Workspace(dev, 512 << 20)
    close16(nchw(ops.bf16_conv2d_fwd(xg, wg.data_ptr(), K, R, R, st, pad)), y.detach())
    close32(nchw(ops.bf16_conv2d_fwd(xg, wg.data_ptr(), K, R, R, st, pad, out16=False)), y.detach())
    if st == 1:
        close16(nchw(ops.bf16_conv2d_dgrad(dyg, wg.data_ptr(), tuple(xg.shape), R, R, st, pad)), x.grad)
        base = r16(torch.randn(xg.shape, generator=g))
        acc = base.cuda().clone()
        ops.bf16_conv2d_dgrad(dyg, wg.data_ptr(), tuple(xg.shape), R, R, st, pad, out=acc, accumulate=True)
        total = x.grad + base.double().permute(0, 3, 1, 2)
        a_, b_ = nchw(acc).double(), total
        bound = (x.grad.abs() + total.abs()) * 2.0 ** -8 + 3e-5 * total.abs().max()   # round(product), then round(sum)
        assert ((a_ - b_).abs() <= bound).all(), ((a_ - b_).abs() - bound).max().item()
    dw = torch.full((K, R, R, C), float("nan"), device=dev)
    ops.bf16_conv2d_wgrad(xg, dyg, dw.data_ptr(), R, R, st, pad, ws)
    close32(nchw(dw), w.grad, 3e-5)
    torch.cuda.synchronize()


def test_bf16_entry_points_reject_unsupported_shapes(dev):
    from deepsense6g_tii_amd import ops
    from deepsense6g_tii_amd._lib import Ds6gError
    x = torch.zeros(8, 48, dtype=BF, device=dev)        # K = 48 is not a multiple of the 64-element k-tile
    w = torch.zeros(64, 48, dtype=BF, device=dev)
    with pytest.raises(Ds6gError):
        ops.bf16_linear_fwd(x, w.data_ptr(), 0, 64)
    dy = torch.zeros(2, 8, 8, 64, dtype=BF, device=dev)
    with pytest.raises(Ds6gError):                       # strided data gradient: not in this kernel
        ops.bf16_conv2d_dgrad(dy, w.data_ptr(), (2, 16, 16, 64), 3, 3, 2, 1)
