"""Thin tensor-level wrappers over the C ABI (include/ds6g.h).

PyTorch is used for device memory and the current HIP stream only; every arithmetic operation is
a kernel in libds6g.so.  Tensors must be fp32, contiguous, on a HIP device; shapes are checked here
on the host because an out-of-bounds kernel can take the whole GPU node down.
"""
from __future__ import annotations

import torch

from ._lib import lib

F32 = torch.float32


def _p(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    """raw hipStream_t of the calling thread's current stream.  torch.cuda.current_stream() builds a Stream object per call
    (~8 us; the step makes ~2000 launches): the two C-level calls below return the same handle in well under a microsecond"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


_STREAM_OBJS = {}


def current_stream_obj():
    """torch.cuda.current_stream() without its ~8 us of device-index plumbing: Stream objects are cached by raw handle (the
    walk switches streams ~400 times per step)"""
    raw = _stream()
    s = _STREAM_OBJS.get(raw)
    if s is None:
        s = torch.cuda.current_stream()
        _STREAM_OBJS[s.cuda_stream] = s
    return s


class on_stream:
    """`with torch.cuda.stream(s)` for the hot path: one C-level set-stream call each way, no Stream objects built"""
    __slots__ = ("s", "prev")

    def __init__(self, s):
        self.s = s
        _STREAM_OBJS.setdefault(s.cuda_stream, s)

    def __enter__(self):
        self.prev = current_stream_obj()
        s = self.s
        torch._C._cuda_setStream(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)
        return s

    def __exit__(self, *exc):
        p = self.prev
        torch._C._cuda_setStream(stream_id=p.stream_id, device_index=p.device_index, device_type=p.device_type)
        return False


def _chk(t, *shape):
    assert t.dtype == F32 and t.is_cuda and t.is_contiguous(), (t.dtype, t.device, t.stride())
    if shape:
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), shape)


class Workspace:
    """One caller-owned scratch buffer shared by all kernels of a stream (split-K slabs, BN / LN /
    column-sum partials).  Stream order makes the reuse safe."""

    def __init__(self, device, nbytes=1 << 30):
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.nbytes = nbytes

    @property
    def ptr(self):
        return self.buf.data_ptr()


# ------------------------------------------------------------------------------------------------
def conv_out_hw(H, W, R, S, stride, pad):
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1


def conv2d_fwd(x, w_ohwi_ptr, K, R, S, stride, pad, out=None):
    N, H, W, C = x.shape
    _chk(x)
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    y = out if out is not None else torch.empty((N, Ho, Wo, K), dtype=F32, device=x.device)
    _chk(y, N, Ho, Wo, K)
    lib().conv2d_fwd(_p(x), w_ohwi_ptr, _p(y), N, H, W, C, K, R, S, stride, pad, _stream())
    return y


def conv2d_bias_act_fwd(x, w_ohwi_ptr, bias_ptr, K, R, S, stride, pad, relu=0, residual=None):
    """inference conv with folded BN: act(conv(x, w) + bias [+ residual]); relu 0 none / 1 before / 2 after the add"""
    N, H, W, C = x.shape
    _chk(x)
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    y = torch.empty((N, Ho, Wo, K), dtype=F32, device=x.device)
    if residual is not None:
        _chk(residual, N, Ho, Wo, K)
    lib().conv2d_bias_act_fwd(_p(x), w_ohwi_ptr, bias_ptr, _p(residual), _p(y), N, H, W, C, K, R, S, stride, pad,
                              int(relu), _stream())
    return y


def bn_fold(w_ohwi_ptr, bn, K, taps, cin, cpad=None):
    """-> (w_folded [K, taps, cpad], bias [K]) of an eval-mode BatchNorm2d folded into the preceding conv"""
    cpad = cin if cpad is None else cpad
    dev = bn.weight.device
    w_out = torch.empty((K, taps, cpad), dtype=F32, device=dev)
    b_out = torch.empty((K,), dtype=F32, device=dev)
    lib().bn_fold(w_ohwi_ptr, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                  bn.running_var.data_ptr(), float(bn.eps), w_out.data_ptr(), b_out.data_ptr(), K, taps, cin, cpad, _stream())
    return w_out, b_out


def winograd_ok(x_shape, K):
    """3x3 / stride 1 / pad 1 conv of an NHWC tensor of this shape to K channels can run as Winograd F(2x2, 3x3)"""
    N, H, W, C = x_shape
    # exact-fp32 products only: the default mode, and f32x6 (fp32-grade: its direct kernels use the six-product split, its
    # 3x3 / stride-1 convs stay on the fp32 Winograd kernel, which is faster than the split direct form)
    return bool(lib().winograd_supported(N, H, W, C, K)) and lib().get_compute_mode() in (0, 3)


def winograd_weights(w_ohwi_ptr, K, C, device, dgrad=False, both=False):
    """U = G g G^T of an OHWI [K, 3, 3, C] filter: [16, K, C] (forward) or, dgrad=True, [16, C, K] of the
    channel-swapped, 180-degree-rotated filter; both=True: (forward, dgrad) from one launch"""
    n = lib().winograd_weight_floats(K, C)
    u = torch.empty(2 * n if both else n, dtype=F32, device=device)
    lib().winograd_weights(w_ohwi_ptr, _p(u), K, C, 2 if both else int(dgrad), _stream())
    return (u[:n], u[n:]) if both else u


def conv3x3_winograd(x, u, K, out=None, accumulate=False):
    """y (+)= conv3x3(x) (stride 1, pad 1) from the transformed filter u [16, K, C]"""
    N, H, W, C = x.shape
    _chk(x)
    y = out if out is not None else torch.empty((N, H, W, K), dtype=F32, device=x.device)
    _chk(y, N, H, W, K)
    lib().conv3x3_winograd_fwd(_p(x), _p(u), _p(y), N, H, W, C, K, int(accumulate), _stream())
    return y


def conv3x3_winograd_bias_act(x, u, bias_ptr, K, relu=0, residual=None):
    """inference 3x3 conv with folded BN in the Winograd domain: act(conv(x) + bias [+ residual])"""
    N, H, W, C = x.shape
    _chk(x)
    y = torch.empty((N, H, W, K), dtype=F32, device=x.device)
    if residual is not None:
        _chk(residual, N, H, W, K)
    lib().conv3x3_winograd_bias_act_fwd(_p(x), _p(u), bias_ptr, _p(residual), _p(y), N, H, W, C, K, int(relu), _stream())
    return y


def winograd_wgrad_ok(x_shape, K):
    N, H, W, C = x_shape
    return bool(lib().winograd_wgrad_supported(N, H, W, C, K)) and lib().get_compute_mode() in (0, 3)


def conv3x3_winograd_wgrad(x, dy, dw_ohwi_ptr, ws: Workspace, accumulate=False):
    """dw [K, 3, 3, C] (+)= weight gradient of the 3x3 / stride 1 / pad 1 conv, computed in the Winograd domain"""
    N, H, W, C = x.shape
    K = dy.shape[3]
    _chk(x)
    _chk(dy, N, H, W, K)
    lib().conv3x3_winograd_wgrad(_p(x), _p(dy), dw_ohwi_ptr, N, H, W, C, K, int(accumulate), ws.ptr, ws.nbytes, _stream())


def conv2d_dgrad(dy, w_ohwi_ptr, x_shape, R, S, stride, pad, out=None, accumulate=False):
    N, H, W, C = x_shape
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    K = dy.shape[3]
    _chk(dy, N, Ho, Wo, K)
    dx = out if out is not None else torch.empty(x_shape, dtype=F32, device=dy.device)
    _chk(dx, *x_shape)
    lib().conv2d_dgrad(_p(dy), w_ohwi_ptr, _p(dx), N, H, W, C, K, R, S, stride, pad, int(accumulate), _stream())
    return dx


def conv2d_wgrad(x, dy, dw_ptr, R, S, stride, pad, ws: Workspace, accumulate=False):
    N, H, W, C = x.shape
    _chk(x)
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    K = dy.shape[3]
    _chk(dy, N, Ho, Wo, K)
    lib().conv2d_wgrad(_p(x), _p(dy), dw_ptr, N, H, W, C, K, R, S, stride, pad, int(accumulate), ws.ptr, ws.nbytes,
                       _stream())


def linear_fwd(x, w_ptr, b_ptr, N, relu=False, residual=None, drop_p=0.0, seed=0, seed_off=0, out=None):
    M, K = x.shape
    _chk(x)
    y = out if out is not None else torch.empty((M, N), dtype=F32, device=x.device)
    _chk(y, M, N)
    if residual is not None:
        _chk(residual, M, N)
    lib().linear_fwd(_p(x), w_ptr, b_ptr, _p(y), M, N, K, int(relu), _p(residual), float(drop_p), seed, seed_off,
                     _stream())
    return y


def linear_dgrad(dy, w_ptr, K, relu_mask_src=None, out=None, accumulate=False):
    M, N = dy.shape
    _chk(dy)
    dx = out if out is not None else torch.empty((M, K), dtype=F32, device=dy.device)
    _chk(dx, M, K)
    if relu_mask_src is not None:
        _chk(relu_mask_src, M, K)
    lib().linear_dgrad(_p(dy), w_ptr, _p(dx), M, N, K, _p(relu_mask_src), int(accumulate), _stream())
    return dx


def linear_wgrad(x, dy, dw_ptr, ws: Workspace, accumulate=False, dbias_ptr=0):
    """dw (+)= dy^T x and, when dbias_ptr is given, dbias (+)= column sums of dy from the same kernel."""
    M, K = x.shape
    M2, N = dy.shape
    assert M == M2
    _chk(x)
    _chk(dy)
    lib().linear_wgrad(_p(x), _p(dy), dw_ptr, dbias_ptr, M, N, K, int(accumulate), ws.ptr, ws.nbytes, _stream())


def colsum(x, out_ptr, ws: Workspace, accumulate=False):
    M, C = x.shape
    _chk(x)
    lib().colsum(_p(x), M, C, out_ptr, int(accumulate), ws.ptr, ws.nbytes, _stream())


# ------------------------------------------------------------------------------------------------
def bn_stats(x2d_rows, C, x, mean, invstd, rm_ptr, rv_ptr, ws: Workspace, eps=1e-5, momentum=0.1):
    lib().bn_stats(_p(x), x2d_rows, C, eps, momentum, _p(mean), _p(invstd), rm_ptr, rv_ptr, ws.ptr, ws.nbytes,
                   _stream())


def bn_eval_prepare(rm_ptr, rv_ptr, C, mean, invstd, eps=1e-5):
    lib().bn_eval_prepare(rm_ptr, rv_ptr, C, eps, _p(mean), _p(invstd), _stream())


def bn_apply(x, mean, invstd, gamma_ptr, beta_ptr, relu, residual=None, out=None):
    _chk(x)
    C = x.shape[-1]
    M = x.numel() // C
    y = out if out is not None else torch.empty_like(x)
    if residual is not None:
        _chk(residual, *x.shape)
    lib().bn_apply(_p(x), _p(mean), _p(invstd), gamma_ptr, beta_ptr, _p(residual), _p(y), M, C, int(relu), _stream())
    return y


def bn_bwd(dy, y_mask, x, mean, invstd, gamma_ptr, dgamma_ptr, dbeta_ptr, ws: Workspace, want_dres=False,
           accumulate=False, dx_out=None, relu_beta_ptr=0):
    """y_mask: activation tensor whose sign gives the ReLU mask, or None; relu_beta_ptr (with y_mask None): BN -> ReLU
    without a residual in between - the mask is recomputed from x inside the kernels."""
    _chk(dy, *x.shape)
    _chk(x)
    C = x.shape[-1]
    M = x.numel() // C
    dx = dx_out if dx_out is not None else torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    lib().bn_bwd(_p(dy), _p(y_mask), _p(x), _p(mean), _p(invstd), gamma_ptr, relu_beta_ptr, _p(dx), dgamma_ptr, dbeta_ptr,
                 _p(dres), M, C, int(accumulate), ws.ptr, ws.nbytes, _stream())
    return dx, dres


def bn_relu_maxpool(x, mean, invstd, gamma_ptr, beta_ptr):
    """maxpool3x3/2(relu(BN(x))) in one pass (the stem) -> (pooled [N, Ho, Wo, C], argmax index uint8)"""
    N, H, W, C = x.shape
    _chk(x)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, Ho, Wo, C), dtype=F32, device=x.device)
    idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    lib().bn_relu_maxpool3x3s2_fwd(_p(x), _p(mean), _p(invstd), gamma_ptr, beta_ptr, _p(y), _p(idx), N, H, W, C, _stream())
    return y, idx


def bn_bwd_maxpool(dpool, idx, x, mean, invstd, gamma_ptr, beta_ptr, dgamma_ptr, dbeta_ptr, ws: Workspace, accumulate=False):
    """backward of bn_relu_maxpool: dx of the BN input from the gradient of the pooled tensor (ReLU mask recomputed from
    x, pool gradient gathered from (dpool, idx) inside the BN kernels)"""
    N, H, W, C = x.shape
    _chk(x)
    _chk(dpool, N, (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1, C)
    dx = torch.empty_like(x)
    lib().bn_bwd_maxpool(_p(dpool), _p(idx), _p(x), _p(mean), _p(invstd), gamma_ptr, beta_ptr, _p(dx), dgamma_ptr, dbeta_ptr,
                         N, H, W, C, int(accumulate), ws.ptr, ws.nbytes, _stream())
    return dx


def layernorm_fwd(x, gamma_ptr, beta_ptr, eps=1e-5):
    M, C = x.shape
    _chk(x)
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=F32, device=x.device)
    rstd = torch.empty(M, dtype=F32, device=x.device)
    lib().layernorm_fwd(_p(x), gamma_ptr, beta_ptr, _p(y), _p(mean), _p(rstd), M, C, eps, _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma_ptr, dgamma_ptr, dbeta_ptr, ws: Workspace, add=None, accumulate=False,
                  out=None, drop=None):
    """-> dx, or (dx, dropout(dx)) when drop = (p, seed, seed_off) with p > 0 (fused second output)"""
    M, C = x.shape
    _chk(dy, M, C)
    _chk(x)
    dx = out if out is not None else torch.empty_like(x)
    if add is not None:
        _chk(add, M, C)
    dxd = torch.empty_like(x) if (drop is not None and drop[0] > 0) else None
    p, seed, off = drop if dxd is not None else (0.0, 0, 0)
    lib().layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), gamma_ptr, _p(add), _p(dx), dgamma_ptr, dbeta_ptr, M, C,
                        int(accumulate), _p(dxd), float(p), seed, off, ws.ptr, ws.nbytes, _stream())
    return dx if drop is None else (dx, dxd if dxd is not None else dx)


# ------------------------------------------------------------------------------------------------
def _rows(t, M, C):
    """2-D fp32 operand whose rows may be a column block of a wider matrix -> row stride in floats"""
    assert t.dtype == F32 and t.is_cuda and tuple(t.shape) == (M, C) and t.stride(1) == 1 and t.stride(0) % 4 == 0, \
        (t.dtype, tuple(t.shape), t.stride())
    return t.stride(0)


def attention_fwd(q, k, v, B, T, nh, ws: Workspace, drop_p=0.0, seed=0, seed_off=0):
    """q, k, v: [B*T, C] each, contiguous or column blocks of one fused projection output (same row stride)."""
    M, C = q.shape
    assert M == B * T
    ldq = _rows(q, M, C)
    assert _rows(k, M, C) == ldq and _rows(v, M, C) == ldq
    o = torch.empty((M, C), dtype=F32, device=q.device)
    lse = torch.empty((B, nh, T), dtype=F32, device=q.device)
    lib().attention_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), B, T, nh, C // nh, ldq, C, float(drop_p), seed, seed_off,
                        ws.ptr, ws.nbytes, _stream())
    return o, lse


def attention_bwd(q, k, v, o, d_o, lse, B, T, nh, ws: Workspace, drop_p=0.0, seed=0, seed_off=0, out=None):
    """-> (dq, dk, dv); `out` = three [B*T, C] views with one common row stride (column blocks of a fused matrix)."""
    M, C = q.shape
    ldq = _rows(q, M, C)
    assert _rows(k, M, C) == ldq and _rows(v, M, C) == ldq
    for t in (o, d_o):
        _chk(t, M, C)
    _chk(lse, B, nh, T)
    delta = torch.empty_like(lse)
    if out is None:
        out = tuple(torch.empty((M, C), dtype=F32, device=q.device) for _ in range(3))
    dq, dk, dv = out
    ldd = _rows(dq, M, C)
    assert _rows(dk, M, C) == ldd and _rows(dv, M, C) == ldd
    lib().attention_bwd(_p(q), _p(k), _p(v), _p(o), _p(d_o), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), B, T, nh,
                        C // nh, ldq, C, ldd, float(drop_p), seed, seed_off, ws.ptr, ws.nbytes, _stream())
    return dq, dk, dv


# ------------------------------------------------------------------------------------------------
def dropout(src, drop_p, seed, seed_off, out=None):
    _chk(src)
    dst = out if out is not None else torch.empty_like(src)
    lib().dropout(_p(src), _p(dst), src.numel(), float(drop_p), seed, seed_off, _stream())
    return dst


def axpby(a, b, alpha=1.0, beta=1.0, out=None):
    _chk(a)
    o = out if out is not None else torch.empty_like(a)
    lib().axpby(_p(a), _p(b), _p(o), a.numel(), float(alpha), float(beta), _stream())
    return o


_MODES = {"f32": 0, "bf16": 1, "f32x3": 2, "f32x6": 3}


def set_compute_mode(mode: str):
    """Process-wide matrix-core mode of the GEMM-shaped kernels (ds6g_set_compute_mode):
    "f32"   - exact fp32 MFMA, the parity path (default);
    "bf16"  - operands rounded to bf16 on the way into the MFMA, fp32 accumulate and storage (throughput mode; the
              reference has no mixed precision, tolerances for it are declared in tests/test_bf16_gpu.py);
    "f32x3" - split bf16: a*b = hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32 accumulate and storage; relative
              product error <= ~2^-16 (tests/test_bf16_gpu.py holds it to the 1e-3 bar of the exact path)."""
    if mode not in _MODES:
        raise ValueError(f"compute mode must be one of {sorted(_MODES)}, got {mode!r}")
    lib().set_compute_mode(_MODES[mode])


def get_compute_mode() -> str:
    return {v: k for k, v in _MODES.items()}[lib().get_compute_mode()]


# ------------------------------------------------------------------------------------------------
# bf16-stored operands (csrc/bgemm.hip): activations / weight shadow bf16, accumulation fp32
BF16 = torch.bfloat16


def _chk16(t, *shape):
    assert t.dtype == BF16 and t.is_cuda and t.is_contiguous(), (t.dtype, t.device, t.stride())
    if shape:
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), shape)


def bf16_linear_fwd(x, w16_ptr, b_ptr, N, relu=False, residual=None, drop_p=0.0, seed=0, seed_off=0, out16=True):
    """y = residual + dropout(act(x w^T + b)): x [M, K] bf16, w [N, K] bf16; y bf16, or fp32 (always with a residual)"""
    M, K = x.shape
    _chk16(x)
    out16 = bool(out16) and residual is None
    y = torch.empty((M, N), dtype=BF16 if out16 else F32, device=x.device)
    if residual is not None:
        _chk(residual, M, N)
    lib().bf16_linear_fwd(_p(x), w16_ptr, b_ptr, _p(y), int(out16), M, N, K, int(relu), _p(residual), float(drop_p), seed,
                          seed_off, _stream())
    return y


def bf16_linear_dgrad(dy, w16_ptr, K, relu_mask_src=None, out16=True, out=None, accumulate=False):
    """dx (+)= (dy w) * (mask > 0): dy [M, N] bf16, w [N, K] bf16; mask bf16 or fp32 [M, K]"""
    M, N = dy.shape
    _chk16(dy)
    dx = out if out is not None else torch.empty((M, K), dtype=BF16 if out16 else F32, device=dy.device)
    assert dx.dtype == (BF16 if out16 else F32) and tuple(dx.shape) == (M, K) and dx.is_contiguous()
    mask16 = 0
    if relu_mask_src is not None:
        assert tuple(relu_mask_src.shape) == (M, K) and relu_mask_src.is_contiguous() and out16
        mask16 = int(relu_mask_src.dtype == BF16)
    lib().bf16_linear_dgrad(_p(dy), w16_ptr, _p(dx), int(out16), M, N, K, _p(relu_mask_src), mask16, int(accumulate), _stream())
    return dx


def bf16_linear_wgrad(x, dy, dw_ptr, ws: Workspace, accumulate=False, dbias_ptr=0):
    """dw (fp32) (+)= dy^T x, dbias (+)= column sums of dy: x [M, K], dy [M, N] bf16"""
    M, K = x.shape
    M2, N = dy.shape
    assert M == M2
    _chk16(x)
    _chk16(dy)
    lib().bf16_linear_wgrad(_p(x), _p(dy), dw_ptr, dbias_ptr, M, N, K, int(accumulate), ws.ptr, ws.nbytes, _stream())


def bf16_conv2d_fwd(x, w16_ptr, K, R, S, stride, pad, out16=True):
    N, H, W, C = x.shape
    _chk16(x)
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    y = torch.empty((N, Ho, Wo, K), dtype=BF16 if out16 else F32, device=x.device)
    lib().bf16_conv2d_fwd(_p(x), w16_ptr, _p(y), int(out16), N, H, W, C, K, R, S, stride, pad, _stream())
    return y


def bf16_conv2d_fwd_bnstats(x, w16_ptr, K, R, S, stride, pad, mean, invstd, rm_ptr, rv_ptr, ws: Workspace, eps=1e-5,
                            momentum=0.1):
    """y = conv(x, w) (bf16) and the train-mode BatchNorm statistics of y from the conv's own epilogue (no pass over y)"""
    N, H, W, C = x.shape
    _chk16(x)
    _chk(mean, K)
    _chk(invstd, K)
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    y = torch.empty((N, Ho, Wo, K), dtype=BF16, device=x.device)
    lib().bf16_conv2d_fwd_bnstats(_p(x), w16_ptr, _p(y), N, H, W, C, K, R, S, stride, pad, eps, momentum, _p(mean),
                                  _p(invstd), rm_ptr, rv_ptr, ws.ptr, ws.nbytes, _stream())
    return y


def bf16_conv2d_dgrad(dy, w16_ptr, x_shape, R, S, stride, pad, out16=True, out=None, accumulate=False):
    N, H, W, C = x_shape
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    K = dy.shape[3]
    _chk16(dy, N, Ho, Wo, K)
    dx = out if out is not None else torch.empty(x_shape, dtype=BF16 if out16 else F32, device=dy.device)
    assert dx.dtype == (BF16 if out16 else F32) and tuple(dx.shape) == tuple(x_shape) and dx.is_contiguous()
    lib().bf16_conv2d_dgrad(_p(dy), w16_ptr, _p(dx), int(out16), N, H, W, C, K, R, S, stride, pad, int(accumulate), _stream())
    return dx


def bf16_conv2d_wgrad(x, dy, dw_ptr, R, S, stride, pad, ws: Workspace, accumulate=False):
    N, H, W, C = x.shape
    _chk16(x)
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    K = dy.shape[3]
    _chk16(dy, N, Ho, Wo, K)
    lib().bf16_conv2d_wgrad(_p(x), _p(dy), dw_ptr, N, H, W, C, K, R, S, stride, pad, int(accumulate), ws.ptr, ws.nbytes, _stream())


def cast_bf16(src, out=None):
    """bf16 copy (RNE) of a contiguous fp32 tensor whose numel is a multiple of 4"""
    _chk(src)
    dst = out if out is not None else torch.empty(src.shape, dtype=BF16, device=src.device)
    assert dst.dtype == BF16 and dst.numel() == src.numel() and dst.is_contiguous()
    lib().cast_f32_bf16(_p(src), _p(dst), src.numel(), _stream())
    return dst


def layernorm_fwd_bf16(x, gamma_ptr, beta_ptr, eps=1e-5):
    """LayerNorm of fp32 rows, output written as bf16 (a GEMM operand) -> (y16, mean, rstd)"""
    M, C = x.shape
    _chk(x)
    y = torch.empty((M, C), dtype=BF16, device=x.device)
    mean = torch.empty(M, dtype=F32, device=x.device)
    rstd = torch.empty(M, dtype=F32, device=x.device)
    lib().layernorm_fwd_bf16out(_p(x), gamma_ptr, beta_ptr, _p(y), _p(mean), _p(rstd), M, C, eps, _stream())
    return y, mean, rstd


def layernorm_bwd_bf16(dy, x, mean, rstd, gamma_ptr, dgamma_ptr, dbeta_ptr, ws: Workspace, add=None, accumulate=False,
                       drop=None, want_drop=True):
    """dy bf16 or fp32 -> (dx fp32, dropout(dx) as bf16); drop = (p, seed, seed_off) or None (= plain bf16 copy of dx);
    want_drop False: no second output (-> (dx, None))"""
    M, C = x.shape
    assert tuple(dy.shape) == (M, C) and dy.is_contiguous() and dy.dtype in (BF16, F32)
    _chk(x)
    if add is not None:
        _chk(add, M, C)
    dx = torch.empty_like(x)
    dxd = torch.empty((M, C), dtype=BF16, device=x.device) if want_drop else None
    p, seed, off = drop if drop is not None else (0.0, 0, 0)
    lib().layernorm_bwd_bf16(_p(dy), int(dy.dtype == BF16), _p(x), _p(mean), _p(rstd), gamma_ptr, _p(add), _p(dx), dgamma_ptr,
                             dbeta_ptr, M, C, int(accumulate), _p(dxd), float(p), seed, off, ws.ptr, ws.nbytes, _stream())
    return dx, dxd


def attention_fwd_bf16out(q, k, v, B, T, nh, ws: Workspace, drop_p=0.0, seed=0, seed_off=0):
    """as attention_fwd, the output written as bf16 (operand of the projection GEMM)"""
    M, C = q.shape
    assert M == B * T
    ldq = _rows(q, M, C)
    assert _rows(k, M, C) == ldq and _rows(v, M, C) == ldq
    o = torch.empty((M, C), dtype=BF16, device=q.device)
    lse = torch.empty((B, nh, T), dtype=F32, device=q.device)
    lib().attention_fwd_bf16out(_p(q), _p(k), _p(v), _p(o), _p(lse), B, T, nh, C // nh, ldq, C, float(drop_p), seed, seed_off,
                                ws.ptr, ws.nbytes, _stream())
    return o, lse


def attention_bwd_bf16(q, k, v, o16, d_o, lse, B, T, nh, ws: Workspace, drop_p=0.0, seed=0, seed_off=0, out=None):
    """o16: the bf16 forward output; d_o fp32; -> dq, dk, dv as bf16 (`out`: three [B*T, C] bf16 views of one row stride)"""
    M, C = q.shape
    ldq = _rows(q, M, C)
    assert _rows(k, M, C) == ldq and _rows(v, M, C) == ldq
    _chk16(o16, M, C)
    _chk(d_o, M, C)
    _chk(lse, B, nh, T)
    assert ws.nbytes >= int(lib().attention_workspace_bytes(B, T, nh, C // nh, C)), "attention_bwd_bf16 needs the hand-over workspace"
    delta = torch.empty_like(lse)
    if out is None:
        out = tuple(torch.empty((M, C), dtype=BF16, device=q.device) for _ in range(3))
    dq, dk, dv = out
    for t in out:
        assert t.dtype == BF16 and tuple(t.shape) == (M, C) and t.stride(1) == 1 and t.stride(0) % 4 == 0
    ldd = dq.stride(0)
    assert dk.stride(0) == ldd and dv.stride(0) == ldd
    lib().attention_bwd_bf16(_p(q), _p(k), _p(v), _p(o16), _p(d_o), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), B, T, nh,
                             C // nh, ldq, C, ldd, float(drop_p), seed, seed_off, ws.ptr, ws.nbytes, _stream())
    return dq, dk, dv


# ---- bf16-storage path: BatchNorm / pooling / resampling on bf16 feature maps (statistics and arithmetic fp32) ----
def bf16_bn_stats(x2d_rows, C, x, mean, invstd, rm_ptr, rv_ptr, ws: Workspace, eps=1e-5, momentum=0.1):
    _chk16(x)
    lib().bf16_bn_stats(_p(x), x2d_rows, C, eps, momentum, _p(mean), _p(invstd), rm_ptr, rv_ptr, ws.ptr, ws.nbytes, _stream())


def bf16_bn_apply(x, mean, invstd, gamma_ptr, beta_ptr, relu, residual=None):
    _chk16(x)
    C = x.shape[-1]
    y = torch.empty_like(x)
    if residual is not None:
        _chk16(residual, *x.shape)
    lib().bf16_bn_apply(_p(x), _p(mean), _p(invstd), gamma_ptr, beta_ptr, _p(residual), _p(y), x.numel() // C, C, int(relu),
                        _stream())
    return y


def bf16_bn_bwd(dy, y_mask, x, mean, invstd, gamma_ptr, dgamma_ptr, dbeta_ptr, ws: Workspace, want_dres=False,
                accumulate=False, relu_beta_ptr=0):
    _chk16(dy, *x.shape)
    _chk16(x)
    if y_mask is not None:
        _chk16(y_mask, *x.shape)
    C = x.shape[-1]
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    lib().bf16_bn_bwd(_p(dy), _p(y_mask), _p(x), _p(mean), _p(invstd), gamma_ptr, relu_beta_ptr, _p(dx), dgamma_ptr, dbeta_ptr,
                      _p(dres), x.numel() // C, C, int(accumulate), ws.ptr, ws.nbytes, _stream())
    return dx, dres


def bn_relu_maxpool_bf16out(x, mean, invstd, gamma_ptr, beta_ptr):
    """the stem's BN -> ReLU -> MaxPool of the fp32 conv output, pooled tensor written as bf16"""
    N, H, W, C = x.shape
    _chk(x)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, Ho, Wo, C), dtype=BF16, device=x.device)
    idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    lib().bn_relu_maxpool3x3s2_fwd_bf16out(_p(x), _p(mean), _p(invstd), gamma_ptr, beta_ptr, _p(y), _p(idx), N, H, W, C, _stream())
    return y, idx


def bn_bwd_maxpool_bf16in(dpool, idx, x, mean, invstd, gamma_ptr, beta_ptr, dgamma_ptr, dbeta_ptr, ws: Workspace, accumulate=False):
    N, H, W, C = x.shape
    _chk(x)
    _chk16(dpool, N, (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1, C)
    dx = torch.empty_like(x)
    lib().bn_bwd_maxpool_bf16in(_p(dpool), _p(idx), _p(x), _p(mean), _p(invstd), gamma_ptr, beta_ptr, _p(dx), dgamma_ptr,
                                dbeta_ptr, N, H, W, C, int(accumulate), ws.ptr, ws.nbytes, _stream())
    return dx


# ---- the 7x7 / 2 stem convolutions of the bf16 configuration (csrc/stem.hip) -------------------------------------------
def bf16_stem_ok(H, W):
    return H % 16 == 0 and W % 32 == 0


def bf16_stem_fwd(x16, w_ohwi_ptr, cin, ws: Workspace, stats=None, rm_ptr=0, rv_ptr=0, eps=1e-5, momentum=0.1):
    """y = conv7x7/2(x) for x [N, H, W, 4] bf16 and the fp32 master filter [64, 7, 7, cin]; stats = (mean, invstd) fp32 [64]
    tensors: also the train-mode BatchNorm statistics of y (running statistics updated in place)"""
    N, H, W, C4 = x16.shape
    _chk16(x16)
    assert C4 == 4 and bf16_stem_ok(H, W) and ws.nbytes >= int(lib().bf16_stem_workspace_bytes())
    y = torch.empty((N, H // 2, W // 2, 64), dtype=BF16, device=x16.device)
    mean, invstd = stats if stats is not None else (None, None)
    lib().bf16_stem_fwd(_p(x16), w_ohwi_ptr, cin, _p(y), N, H, W, eps, momentum, _p(mean), _p(invstd), rm_ptr, rv_ptr, ws.ptr,
                        ws.nbytes, _stream())
    return y


def bf16_stem_wgrad(x16, dy16, dw_ptr, cin, ws: Workspace, accumulate=False):
    N, H, W, C4 = x16.shape
    _chk16(x16)
    _chk16(dy16, N, H // 2, W // 2, 64)
    assert C4 == 4 and bf16_stem_ok(H, W) and ws.nbytes >= int(lib().bf16_stem_workspace_bytes())
    lib().bf16_stem_wgrad(_p(x16), _p(dy16), dw_ptr, cin, N, H, W, int(accumulate), ws.ptr, ws.nbytes, _stream())


def bf16_stem_bn_relu_maxpool(x16, mean, invstd, gamma_ptr, beta_ptr):
    N, H, W, C = x16.shape
    _chk16(x16)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, Ho, Wo, C), dtype=BF16, device=x16.device)
    idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x16.device)
    lib().bf16_stem_bn_relu_maxpool_fwd(_p(x16), _p(mean), _p(invstd), gamma_ptr, beta_ptr, _p(y), _p(idx), N, H, W, C, _stream())
    return y, idx


def bf16_stem_bn_bwd_maxpool(dpool, idx, x16, mean, invstd, gamma_ptr, beta_ptr, dgamma_ptr, dbeta_ptr, ws: Workspace, accumulate=False):
    N, H, W, C = x16.shape
    _chk16(x16)
    _chk16(dpool, N, (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1, C)
    dx = torch.empty_like(x16)
    lib().bf16_stem_bn_bwd_maxpool(_p(dpool), _p(idx), _p(x16), _p(mean), _p(invstd), gamma_ptr, beta_ptr, _p(dx), dgamma_ptr,
                                   dbeta_ptr, N, H, W, C, int(accumulate), ws.ptr, ws.nbytes, _stream())
    return dx


def _rows16(t, M, C):
    assert t.dtype == BF16 and t.is_cuda and tuple(t.shape) == (M, C) and t.stride(1) == 1 and t.stride(0) % 8 == 0, \
        (t.dtype, tuple(t.shape), t.stride())
    return t.stride(0)


def attention_fwd_bf16(q, k, v, B, T, nh, ws: Workspace, drop_p=0.0, seed=0, seed_off=0):
    """all-bf16 attention: q, k, v [B*T, C] bf16 (column blocks of one fused projection output allowed) -> (o bf16, lse fp32)"""
    M, C = q.shape
    assert M == B * T
    ldq = _rows16(q, M, C)
    assert _rows16(k, M, C) == ldq and _rows16(v, M, C) == ldq
    o = torch.empty((M, C), dtype=BF16, device=q.device)
    lse = torch.empty((B, nh, T), dtype=F32, device=q.device)
    lib().attention_fwd_bf16(_p(q), _p(k), _p(v), _p(o), _p(lse), B, T, nh, C // nh, ldq, C, float(drop_p), seed, seed_off,
                             ws.ptr, ws.nbytes, _stream())
    return o, lse


def attention_bwd_bf16io(q, k, v, o, d_o, lse, B, T, nh, ws: Workspace, drop_p=0.0, seed=0, seed_off=0, out=None):
    """all-bf16 backward: q, k, v, o, d_o bf16 -> dq, dk, dv bf16 (`out`: three [B*T, C] bf16 views of one row stride)"""
    M, C = q.shape
    ldq = _rows16(q, M, C)
    assert _rows16(k, M, C) == ldq and _rows16(v, M, C) == ldq
    _chk16(o, M, C)
    _chk16(d_o, M, C)
    _chk(lse, B, nh, T)
    assert ws.nbytes >= int(lib().attention_workspace_bytes(B, T, nh, C // nh, C)), "attention_bwd_bf16io needs the hand-over workspace"
    delta = torch.empty_like(lse)
    if out is None:
        out = tuple(torch.empty((M, C), dtype=BF16, device=q.device) for _ in range(3))
    dq, dk, dv = out
    for t in out:
        assert t.dtype == BF16 and tuple(t.shape) == (M, C) and t.stride(1) == 1 and t.stride(0) % 4 == 0
    ldd = dq.stride(0)
    assert dk.stride(0) == ldd and dv.stride(0) == ldd
    lib().attention_bwd_bf16io(_p(q), _p(k), _p(v), _p(o), _p(d_o), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), B, T, nh,
                               C // nh, ldq, C, ldd, float(drop_p), seed, seed_off, ws.ptr, ws.nbytes, _stream())
    return dq, dk, dv
