"""Parity at the shapes the benchmark actually runs.

bench.py times bs=12 -> N = 60 frames per trunk, B*T = 11 544 token rows; the kernels pick tiles, split-K factors and
attention key-splits by size, so the small-shape parity tests do not reach the code paths the headline number runs.  Here:
  * every matrix kernel alone at N = 60 / B = 12 / M = 11 544 against torch fp32 on the CPU (same tolerances as the
    small-shape tests in test_ops_gpu.py);
  * the full path at bs = 12, seq 5, n_layer 8 against the CPU oracle (logits / loss 1e-3 = the north-star bar; gradient
    tensors spread over join / GPT4 / layer4 / GPT1 / stems against one oracle backward);
  * BASELINE configs[1] (zeroed LiDAR / radar = the reference's "zerolike" missing-modality semantics,
    /root/reference/mambafuser_seq.py:384-391) in exact fp32 and in bf16 mode;
  * the single-GPU slice of BASELINE configs[4] (bs = 32, EMA, bf16 mode, cosine LR);
  * model-level dropout: the oracle run on the very masks the HIP path drew (rebuilt on the CPU from the counter hash).
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def l2rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


def close(a, b, rtol, atol):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"max err {err:.3e} vs ref max {ref:.3e}"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.cpu().permute(0, 3, 1, 2).contiguous()


# ---------------------------------------------------------------------------------------------------------------------
# CPU restatement of the counter-hash dropout of csrc/common.h (ds6g_hash32 / ds6g_rand_u32 / ds6g_keep)
def _hash32(x):
    """ds6g_hash32 on a uint32 array (numpy uint32 arithmetic wraps mod 2^32 like the device code)"""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def _keep_bits(seed, off, numel, p):
    """keep(idx) for the linear element indices off .. off + numel - 1:
    keep(idx) = hash32(lo(idx) ^ key ^ hi(idx) * 0x9E3779B9) >= floor(p * 2^32), key = lo(seed) ^ hi(seed) * 0x85ebca6b
    -> bool array.  Generated in slabs so that a 44 M element attention mask needs no multi-GB temporaries."""
    key = np.uint32((seed & 0xffffffff) ^ (((seed >> 32) * 0x85ebca6b) & 0xffffffff))
    thr = np.uint32(int(float(np.float32(p)) * 4294967296.0))
    out = np.empty(numel, dtype=np.bool_)
    SLAB = 1 << 22

    def slab(s0):
        n = min(SLAB, numel - s0)
        idx = np.arange(off + s0, off + s0 + n, dtype=np.uint64)
        lo = (idx & np.uint64(0xffffffff)).astype(np.uint32)
        hi = (idx >> np.uint64(32)).astype(np.uint32)
        hi *= np.uint32(0x9E3779B9)
        lo ^= key
        lo ^= hi
        out[s0:s0 + n] = _hash32(lo) >= thr

    starts = range(0, numel, SLAB)
    if numel > SLAB:      # numpy releases the GIL inside these array ops
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(_threads()) as ex:
            list(ex.map(slab, starts))
    else:
        for s0 in starts:
            slab(s0)
    return out


def _keep_bits_attn(seed, off, shape, p):
    """attn_drop masks (csrc/common.h Ds6gKeep4Base / ds6g_keep4): FOUR decisions per hash.  Element (row, key) of the
    [B * nh * T][T] probability matrix belongs to key quad  row * ceil(T / 4) + (key >> 2)  (counter = off + that index);
    the quad's two words w0 = fin(x * 0x846ca68b), w1 = fin(x * 0xC2B2AE35) (x = two finalizer rounds on the keyed counter)
    hold the decisions of keys 4k .. 4k + 3 in their 16-bit halves (low w0, high w0, low w1, high w1); kept when the half
    >= floor(p * 2^16)."""
    B, nh, T, T2 = shape
    assert T == T2
    Tq4 = (T + 3) // 4
    rows = B * nh * T
    key = np.uint32((seed & 0xffffffff) ^ (((seed >> 32) * 0x85ebca6b) & 0xffffffff))
    thr16 = np.uint32(int(float(np.float32(p)) * 4294967296.0) >> 16)
    out = np.empty((rows, Tq4 * 4), dtype=np.bool_)
    RS = 1 << 14    # rows per slab

    def slab(r0):
        r1 = min(rows, r0 + RS)
        idx = np.arange(off + r0 * Tq4, off + r1 * Tq4, dtype=np.uint64)
        x = (idx & np.uint64(0xffffffff)).astype(np.uint32)
        hi = (idx >> np.uint64(32)).astype(np.uint32)
        hi *= np.uint32(0x9E3779B9)
        x ^= key
        x ^= hi
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7feb352d)
        x ^= x >> np.uint32(15)
        w0 = x * np.uint32(0x846ca68b)
        w0 ^= w0 >> np.uint32(16)
        w1 = x * np.uint32(0xC2B2AE35)
        w1 ^= w1 >> np.uint32(16)
        o = out[r0:r1].reshape(-1, 4)
        o[:, 0] = (w0 & np.uint32(0xffff)) >= thr16
        o[:, 1] = (w0 >> np.uint32(16)) >= thr16
        o[:, 2] = (w1 & np.uint32(0xffff)) >= thr16
        o[:, 3] = (w1 >> np.uint32(16)) >= thr16

    starts = range(0, rows, RS)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(_threads()) as ex:
        list(ex.map(slab, starts))
    return np.ascontiguousarray(out[:, :T]).reshape(B, nh, T, T)


def _keep_bits_site(seed, off, shape, p):
    """the keep bits of one dropout site: attention-probability sites ((B, nh, T, T)) draw four decisions per hash, every
    other site (embd_drop / resid_drop on (B, T, C)) one per element"""
    if len(shape) == 4:
        return _keep_bits_attn(seed, off, tuple(shape), p)
    return _keep_bits(seed, off, int(np.prod(shape)), p).reshape(shape)


def _drop_scale(p, shape=None):
    """the kernels' fp32 inverted-dropout scale: 1 / (1 - p) at the one-decision-per-hash sites; at the attention-probability
    sites (4-D shapes: four 16-bit decisions per hash) 1 / (1 - floor(p 2^16) / 2^16), the probability the mask realises
    (csrc/common.h ds6g_attn_drop_params)"""
    if shape is not None and len(shape) == 4:
        t16 = int(float(np.float32(p)) * 4294967296.0) >> 16
        return float(np.float32(1.0) / (np.float32(1.0) - np.float32(t16) * np.float32(1.0 / 65536.0))) if t16 else 1.0
    return float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


def _keep_mask(seed, off, shape, p, dtype=torch.float32):
    """-> keep mask scaled by the kernels' fp32 scale"""
    bits = torch.from_numpy(_keep_bits_site(seed, off, tuple(shape), p))
    return bits.to(dtype) * _drop_scale(p, shape)


# ---------------------------------------------------------------------------------------------------------------------
# conv layers of the trunks at N = 60 (bs = 12 x 5 frames): N, H, W, C, K, R, stride, pad
CONV60 = [
    (60, 64, 64, 64, 64, 3, 1, 1),      # layer1 3x3
    (60, 32, 32, 128, 128, 3, 1, 1),    # layer2 3x3
    (60, 16, 16, 256, 256, 3, 1, 1),    # layer3 3x3
    (60, 8, 8, 512, 512, 3, 1, 1),      # layer4 3x3
    (60, 64, 64, 64, 128, 3, 2, 1),     # layer2.0.conv1 (stride 2)
    (60, 32, 32, 128, 256, 3, 2, 1),    # layer3.0.conv1
    (60, 16, 16, 256, 512, 3, 2, 1),    # layer4.0.conv1
    (60, 64, 64, 64, 128, 1, 2, 0),     # layer2.0.downsample
    (60, 16, 16, 256, 512, 1, 2, 0),    # layer4.0.downsample
    (60, 256, 256, 4, 64, 7, 2, 3),     # stem 7x7/2 on the channel-padded input
]


@pytest.mark.parametrize("case", CONV60, ids=lambda c: "x".join(map(str, c)))
def test_conv_kernels_at_bench_batch(dev, case):
    """direct implicit GEMM (fwd / dgrad / wgrad, incl. its split-K choice at this size) and, for the 3x3 / stride-1
    layers, the Winograd fwd / dgrad / wgrad kernels - all against torch conv2d autograd on the CPU"""
    from deepsense6g_tii_amd import ops
    N, H, W, C, K, R, st, pad = case
    torch.set_num_threads(_threads())
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, R, R, generator=g) / math.sqrt(C * R * R)
    if C == 4:          # stem: the 4th input channel is the zero padding of a 3-channel image
        x[:, 3] = 0
    x.requires_grad_(True)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, st, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ws = ops.Workspace(dev, 1 << 30)
    xg, wg, dyg = nhwc(x.detach()), w.detach().permute(0, 2, 3, 1).contiguous().cuda(), nhwc(dy)
    close(nchw(ops.conv2d_fwd(xg, wg.data_ptr(), K, R, R, st, pad)), y.detach(), 2e-5, 2e-5)
    if C != 4:          # the stem has no data gradient on the path (its input is the data)
        close(nchw(ops.conv2d_dgrad(dyg, wg.data_ptr(), tuple(xg.shape), R, R, st, pad)), x.grad, 2e-5, 2e-5)
    dwg = torch.full_like(wg, float("nan"))
    ops.conv2d_wgrad(xg, dyg, dwg.data_ptr(), R, R, st, pad, ws)
    # a weight gradient at N = 60 sums 60*Ho*Wo products per entry: same relative bar, measured against its own scale
    close(dwg.cpu().permute(0, 3, 1, 2), w.grad, 1e-4, 1e-4)
    if R == 3 and st == 1:
        assert ops.winograd_ok(xg.shape, K) and ops.winograd_ok(dyg.shape, C)
        u, ud = ops.winograd_weights(wg.data_ptr(), K, C, dev, both=True)
        close(nchw(ops.conv3x3_winograd(xg, u, K)), y.detach(), 2e-5, 2e-5)
        close(nchw(ops.conv3x3_winograd(dyg, ud, C)), x.grad, 2e-5, 2e-5)
        base = torch.randn(xg.shape, generator=g).cuda()
        acc = ops.conv3x3_winograd(dyg, ud, C, out=base.clone(), accumulate=True)
        close(nchw(acc - base), x.grad, 2e-5, 2e-5)
        if ops.winograd_wgrad_ok(xg.shape, K):
            dw2 = torch.full_like(wg, float("nan"))
            ops.conv3x3_winograd_wgrad(xg, dyg, dw2.data_ptr(), ws)
            close(dw2.cpu().permute(0, 3, 1, 2), w.grad, 1e-4, 1e-4)
    torch.cuda.synchronize()


# GPT linears at M = B*T = 12 * 962: (N, K) of the fused q|k|v projection of stage 1, and the stage-4 MLP pair
@pytest.mark.parametrize("N,K", [(192, 64), (2048, 512), (512, 2048), (1536, 512), (64, 256)])
def test_linear_kernels_at_bench_rows(dev, N, K):
    from deepsense6g_tii_amd import ops
    M = 12 * 962
    torch.set_num_threads(_threads())
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(M, K, generator=g, requires_grad=True)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).requires_grad_(True)
    b = torch.randn(N, generator=g, requires_grad=True)
    res = torch.randn(M, N, generator=g)
    lin = F.linear(x, w, b)
    y = F.relu(lin) + res
    dy = torch.randn(M, N, generator=g)
    y.backward(dy)
    ws = ops.Workspace(dev, 1 << 30)
    xg, wg, bg = x.detach().cuda(), w.detach().cuda(), b.detach().cuda()
    close(ops.linear_fwd(xg, wg.data_ptr(), bg.data_ptr(), N, relu=True, residual=res.cuda()), y.detach(), 2e-5, 2e-5)
    dmask = (dy * (lin.detach() > 0)).cuda()      # the CPU's ReLU decisions, so both backward passes see one mask
    close(ops.linear_dgrad(dmask, wg.data_ptr(), K), x.grad, 2e-5, 2e-5)
    dwg, dbg = torch.full_like(wg, float("nan")), torch.full_like(bg, float("nan"))
    ops.linear_wgrad(xg, dmask, dwg.data_ptr(), ws, dbias_ptr=dbg.data_ptr())
    close(dwg, w.grad, 1e-4, 1e-4)
    close(dbg, b.grad, 1e-4, 1e-4)
    torch.cuda.synchronize()


@pytest.mark.parametrize("pdrop", [0.0, 0.1], ids=["nodrop", "drop0.1"])
@pytest.mark.parametrize("hd", [16, 32, 64, 128])
def test_attention_at_bench_batch(dev, hd, pdrop):
    _attention_case(dev, hd, pdrop, B=12, T=962)


@pytest.mark.parametrize("hd,pdrop,bf16", [(16, 0.1, False), (32, 0.0, False), (64, 0.1, False), (128, 0.1, False),
                                           (64, 0.1, True), (128, 0.1, True)])
def test_attention_at_1922_tokens(dev, hd, pdrop, bf16):
    """the 30 -> 5 variant's sequence (seq_len 10 => T = 1922 = 60 * 32 + 2 tokens, /root/reference/model2_seq_30to5.py:188)
    at the kernel level, B = 3: forward and every backward form against torch autograd with CPU-rebuilt masks (fp32
    storage), and the bf16-stored kernels against the same reference on bf16-rounded operands."""
    if bf16:
        _attention_case_bf16(dev, hd, pdrop, B=3, T=1922)
    else:
        _attention_case(dev, hd, pdrop, B=3, T=1922)


def _attention_case_bf16(dev, hd, pdrop, B, T):
    from deepsense6g_tii_amd import ops
    from deepsense6g_tii_amd._lib import lib
    nh = 4
    C = nh * hd
    BF = torch.bfloat16
    torch.set_num_threads(_threads())
    g = torch.Generator().manual_seed(hd + 1)
    q, k, v = (torch.randn(B * T, C, generator=g).to(BF).double().requires_grad_(True) for _ in range(3))
    seed, off = 0x5DEECE66D ^ (hd << 33), (5 << 40) + 77 * 1024

    def heads(t):
        return t.view(B, T, nh, hd).transpose(1, 2)

    att = torch.softmax((heads(q) @ heads(k).transpose(-2, -1)) * (1.0 / math.sqrt(hd)), dim=-1)
    if pdrop > 0:
        att = att * _keep_mask(seed, off, (B, nh, T, T), pdrop, dtype=torch.float64)
    o = (att @ heads(v)).transpose(1, 2).reshape(B * T, C)
    do = torch.randn(B * T, C, generator=g).to(BF).double()
    o.backward(do)
    dev16 = lambda t: t.detach().to(BF).cuda()  # noqa: E731
    qg, kg, vg, dog = dev16(q), dev16(k), dev16(v), dev16(do)
    ws = ops.Workspace(dev, max(int(lib().attention_workspace_bytes(B, T, nh, hd, C)), 256 << 20))
    kw = dict(drop_p=pdrop, seed=seed, seed_off=off) if pdrop > 0 else {}
    og, lse = ops.attention_fwd_bf16(qg, kg, vg, B, T, nh, ws, *((pdrop, seed, off) if pdrop > 0 else ()))
    bar = lambda a, b: (a.double().cpu() - b).abs().max().item() / b.abs().max().item()  # noqa: E731
    assert bar(og, o.detach()) < 2e-2   # builder-declared bar of the bf16-stored kernels (tests/test_bgemm_gpu.py)
    dq, dk, dv = ops.attention_bwd_bf16io(qg, kg, vg, og, dog, lse, B, T, nh, ws, **kw)
    for got, want, name in ((dq, q.grad, "dq"), (dk, k.grad, "dk"), (dv, v.grad, "dv")):
        assert bar(got, want) < 2e-2, (name, bar(got, want))
    torch.cuda.synchronize()


def _attention_case(dev, hd, pdrop, B, T):
    """B = 12, T = 962, 4 heads: forward, and both backward forms (dS / P hand-over = 5 products, and the recomputing
    form) against torch autograd on the materialised scores.  pdrop = 0.1 is what bench.py times (attn_pdrop of
    /root/reference/model2_seq.py:103-105): the keep mask of all B * nh * T * T probabilities is rebuilt on the CPU from
    the counter hash and applied to the materialised softmax, so the per-split counter offsets of the key-split forward,
    of the split dK/dV kernel with its dS / dropped-P hand-over and of the recomputing dQ kernel are all checked at the
    size the headline runs (1536 workgroups).  The counter offset carries a salt above 2^40 plus a site offset that is not
    a multiple of the row length, like a mid-forward site of the model."""
    from deepsense6g_tii_amd import ops
    from deepsense6g_tii_amd._lib import lib
    nh = 4
    C = nh * hd
    torch.set_num_threads(_threads())
    g = torch.Generator().manual_seed(hd)
    q, k, v = (torch.randn(B * T, C, generator=g, requires_grad=True) for _ in range(3))
    seed, off = 0x5DEECE66D ^ (hd << 33), (3 << 40) + 177 * 1024

    def heads(t):
        return t.view(B, T, nh, hd).transpose(1, 2)

    att = torch.softmax((heads(q) @ heads(k).transpose(-2, -1)) * (1.0 / math.sqrt(hd)), dim=-1)
    if pdrop > 0:
        att = att * _keep_mask(seed, off, (B, nh, T, T), pdrop)
    o = (att @ heads(v)).transpose(1, 2).reshape(B * T, C)
    do = torch.randn(B * T, C, generator=g)
    o.backward(do)
    qg, kg, vg, dog = q.detach().cuda(), k.detach().cuda(), v.detach().cuda(), do.cuda()
    need = int(lib().attention_workspace_bytes(B, T, nh, hd, C))
    ws = ops.Workspace(dev, max(need, 256 << 20))
    kw = dict(drop_p=pdrop, seed=seed, seed_off=off) if pdrop > 0 else {}
    og, lse = ops.attention_fwd(qg, kg, vg, B, T, nh, ws, **kw)
    close(og, o.detach(), 2e-5, 2e-5)
    hand = ops.attention_bwd(qg, kg, vg, og, dog, lse, B, T, nh, ws, **kw)
    lib().set_debug_flags(0x01000000)      # force the recomputing form
    try:
        reco = ops.attention_bwd(qg, kg, vg, og, dog, lse, B, T, nh, ws, **kw)
    finally:
        lib().set_debug_flags(0)
    forms = [hand, reco]
    if hd >= 128:                          # hd 128: the default fuses dK + dV; also the dK kernel + dropped-P tiles + dV kernel form
        lib().set_debug_flags(0x04000000)
        try:
            forms.append(ops.attention_bwd(qg, kg, vg, og, dog, lse, B, T, nh, ws, **kw))
        finally:
            lib().set_debug_flags(0)
    for got in forms:
        close(got[0], q.grad, 1e-4, 2e-5)
        close(got[1], k.grad, 1e-4, 2e-5)
        close(got[2], v.grad, 1e-4, 2e-5)
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------------------------------
def _build(dev, kw, seed):
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from oracle import fusion_ref as fr
    rcfg = fr.RefConfig(**kw)
    sd = fr.make_state(rcfg, seed=seed)
    model = TransFuser(GlobalConfig(**kw), dev)
    model.load_state_dict(sd, strict=True)
    return model, rcfg, sd


GRAD_PROBES = (
    "join.0.weight", "join.4.bias",
    "encoder.transformer4.blocks.7.mlp.2.weight", "encoder.transformer4.blocks.0.attn.query.weight",
    "encoder.transformer4.pos_emb", "encoder.vel_emb4.weight",
    "encoder.image_encoder.features.layer4.2.conv2.weight", "encoder.radar_encoder._model.layer4.0.downsample.0.weight",
    "encoder.lidar_encoder._model.layer3.1.bn2.weight",
    "encoder.transformer1.blocks.3.attn.value.weight", "encoder.transformer1.blocks.0.ln1.bias",
    "encoder.image_encoder.features.layer1.0.conv1.weight",
    "encoder.image_encoder.features.conv1.weight", "encoder.lidar_encoder._model.conv1.weight",
    "encoder.radar_encoder._model.bn1.bias",
)


class _MaskBook:
    """mask_fn for the oracle (oracle/fusion_ref.py Ctx): hands out, site by site in forward order, the keep masks the HIP
    path drew - rebuilt on the CPU from (seed, salt + site offset), each site advanced to the next multiple of 1024 like
    TransFuser._next_drop.  The bits are cached (1 byte per element: 1.4 GB for a bs = 12 / n_layer = 8 forward), so the
    fp32 and the fp64 oracle runs see the same masks without hashing 1.4 G counters twice."""

    def __init__(self, seed, salt):
        self.seed, self.salt, self.cache = seed, salt, {}
        self.rewind()

    def rewind(self):
        self.counter = self.salt

    def __call__(self, shape, p):
        n = int(np.prod(shape))
        off = self.counter
        self.counter += (n + 1023) // 1024 * 1024
        bits = self.cache.get(off)
        if bits is None:
            bits = self.cache[off] = torch.from_numpy(_keep_bits_site(self.seed, off, tuple(shape), p))
        assert bits.shape == tuple(shape)
        return bits, _drop_scale(p, shape)


class _MaskFn:
    """adapter: the oracle multiplies by what mask_fn returns; build the scaled mask in the dtype of the run"""

    def __init__(self, book, dtype):
        self.book, self.dtype = book, dtype
        book.rewind()

    def __call__(self, shape, p):
        bits, scale = self.book(shape, p)
        return bits.to(self.dtype) * scale


def _focal64(lg, t):
    """sigmoid focal loss (alpha .25, gamma 2, mean) in the dtype of lg - oracle/train_ref.py:20-38 without the fp32 cast"""
    p = torch.sigmoid(lg)
    ce = F.binary_cross_entropy_with_logits(lg, t, reduction="none")
    pt = p * t + (1 - p) * (1 - t)
    return ((0.25 * t + 0.75 * (1 - t)) * ce * (1 - pt) ** 2).mean()


def _b12_ctx(st, dtype, capture=None):
    from oracle import fusion_ref as fr
    if not st["dropout"]:
        return fr.Ctx(training=True, capture=capture)
    return fr.Ctx(training=True, dropout=True, capture=capture, mask_fn=_MaskFn(st["book"], dtype))


def _b12_hip_and_oracle32(dev, dropout):
    """ONE forward + backward of the HIP path at the benchmark's shape (bs 12, seq 5, n_layer 8) and of the fp32 CPU
    oracle on the same weights and inputs.  dropout = False: all three pdrop = 0.  dropout = True: the configuration
    bench.py times (embd / attn / resid pdrop 0.1 of /root/reference/config_seq.py:33-35) - the oracle runs on the masks
    the HIP path drew, rebuilt on the CPU."""
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    kw = {} if dropout else dict(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    model, rcfg, sd = _build(dev, kw, seed=12 + dropout)
    inputs = fr.make_inputs(rcfg, 12, seed=112 + dropout)[:5]
    imgs, lids, rads, gps, target = inputs
    model.train()
    cap = {}
    model._capture = cap
    loss, logits = model.train_step_loss(imgs, lids, rads, gps, target)
    model._capture = None
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
    caps = {k: ([t.cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in cap.items()}
    st = dict(dropout=dropout, rcfg=rcfg, sd=sd, inputs=inputs, loss=float(loss), logits=logits.cpu(), grads=grads, cap=caps)
    if dropout:
        assert model._salt_host == int(model._salt.item()) == model.SALT_STRIDE
        st["book"] = _MaskBook(model._seed, model._salt_host)
        st["sites_counted"] = model._drop_counter
    del model, cap
    torch.cuda.empty_cache()
    torch.set_num_threads(_threads())
    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    ocap = {}
    ologits = fr.transfuser_forward(sdo, imgs, lids, rads, gps, rcfg, _b12_ctx(st, torch.float32, ocap))
    if dropout:   # the oracle visited exactly the sites the HIP walk counted
        book = st["book"]
        assert book.counter - book.salt == st["sites_counted"], (book.counter - book.salt, st["sites_counted"])
    oloss = tr.sigmoid_focal_loss(ologits, target)
    oloss.backward()
    st.update(ologits=ologits.detach(), oloss=float(oloss.detach()),
              ograds={n: v.grad for n, v in sdo.items() if isinstance(v, torch.Tensor) and v.requires_grad},
              ocap={k: ([t.detach() for t in v] if isinstance(v, (list, tuple)) else v.detach()) for k, v in ocap.items()})
    return st


def _b12_oracle64(st):
    """fp64 run of the oracle on the same weights / inputs / masks: the yardstick for how far two correct fp32 backward
    passes of this network differ (its own fixture, so the test runner reports progress between the runs)"""
    from oracle import fusion_ref as fr
    if "grads64" in st:
        return st
    torch.set_num_threads(_threads())
    dt = torch.float64
    sd64 = {k: (v.to(dt).clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else
                (v.to(dt) if v.is_floating_point() else v.clone())) for k, v in st["sd"].items()}
    imgs, lids, rads, gps, target = st["inputs"]
    cast = lambda seq: [t.to(dt) for t in seq]  # noqa: E731
    lg = fr.transfuser_forward(sd64, cast(imgs), cast(lids), cast(rads), gps.to(dt), st["rcfg"], _b12_ctx(st, dt))
    _focal64(lg, target.double()).backward()
    st.update(logits64=lg.detach(), grads64={n: v.grad for n, v in sd64.items() if isinstance(v, torch.Tensor) and v.requires_grad})
    return st


@pytest.fixture(scope="module")
def run_b12(dev):
    return _b12_hip_and_oracle32(dev, dropout=False)


@pytest.fixture(scope="module")
def run_b12_f64(run_b12):
    return _b12_oracle64(run_b12)


@pytest.fixture(scope="module")
def run_b12_drop(dev):
    return _b12_hip_and_oracle32(dev, dropout=True)


@pytest.fixture(scope="module")
def run_b12_drop_f64(run_b12_drop):
    return _b12_oracle64(run_b12_drop)


def _forward_report(r):
    report = []
    for name in ("stem", "layer1", "layer2", "layer3", "layer4"):
        for m in range(3):
            report.append((f"{name}[{m}]", rel(nchw(r["cap"][name][m]), r["ocap"][name][m])))
    report.append(("fused", rel(r["cap"]["fused"], r["ocap"]["fused"])))
    report.append(("logits", rel(r["logits"], r["ologits"])))
    msg = "\n".join(f"{k:12s} {v:.3e}" for k, v in report)
    print(msg)
    assert max(v for _, v in report) < TOL, msg
    assert abs(r["loss"] - r["oloss"]) < TOL * abs(r["oloss"]), (r["loss"], r["oloss"])


def _probe_gradients(r, bar):
    rows = [(n, l2rel(r["grads"][n], r["ograds"][n])) for n in GRAD_PROBES]
    msg = "\n".join(f"{n:70s} {e:.3e}" for n, e in rows)
    print(msg)
    assert max(e for _, e in rows) < bar, msg
    # and no systematic scale error: the norms agree much tighter than the entries
    for n in GRAD_PROBES:
        a, b = r["grads"][n].double().norm().item(), r["ograds"][n].double().norm().item()
        assert abs(a - b) < 5e-3 * b, (n, a, b)


def _all_gradients_vs_fp64(r):
    """EVERY parameter gradient of the bs = 12 step (~490 tensors) against the fp64 oracle run, with the fp32 CPU oracle
    held to the same yardstick: gradients of this network are ill-conditioned in fp32 (ReLU / max-pool decisions flip on
    rounding), so 'how far may a correct fp32 backward be from fp64' is measured, not assumed.  Two error measures per
    tensor: max-abs over the tensor's largest entry, and L2-relative.  Bars: HIP median <= 3x the oracle's median; HIP
    worst tensor <= 3x the oracle's worst (or 5 %); no tensor off by more than 15 % (a wrong formula, a missing term or a
    wrong dropout / split offset gives O(1) on the tensors it touches)."""
    hip, o32, hip_l2, o32_l2 = [], [], [], []
    for k, ref in r["grads64"].items():
        scale = ref.abs().max().item()
        g = r["grads"][k].double()
        if scale < 1e-12:  # attn.key.bias: exactly-zero gradient (softmax is invariant to a per-query constant)
            assert g.abs().max().item() < 1e-7, k
            continue
        hip.append(((g - ref).abs().max().item() / scale, k))
        o32.append(((r["ograds"][k].double() - ref).abs().max().item() / scale, k))
        hip_l2.append((l2rel(g, ref), k))
        o32_l2.append((l2rel(r["ograds"][k], ref), k))
    assert len(hip) > 400, len(hip)
    for v in (hip, o32, hip_l2, o32_l2):
        v.sort(reverse=True)
    med = lambda v: v[len(v) // 2][0]  # noqa: E731
    print("grad err vs fp64 over %d tensors (max-abs / largest entry): HIP median %.3e max %.3e (%s) | oracle32 median %.3e "
          "max %.3e (%s)" % (len(hip), med(hip), hip[0][0], hip[0][1], med(o32), o32[0][0], o32[0][1]), flush=True)
    print("                                  (L2-relative)            : HIP median %.3e max %.3e (%s) | oracle32 median %.3e "
          "max %.3e (%s)" % (med(hip_l2), hip_l2[0][0], hip_l2[0][1], med(o32_l2), o32_l2[0][0], o32_l2[0][1]), flush=True)
    assert med(hip) < 3 * med(o32) + 1e-4
    assert hip[0][0] < max(3 * o32[0][0], 0.05), hip[:5]
    assert hip[0][0] < 0.15, hip[:5]
    assert med(hip_l2) < 3 * med(o32_l2) + 1e-4
    assert hip_l2[0][0] < max(3 * o32_l2[0][0], 0.05), hip_l2[:5]
    assert hip_l2[0][0] < 0.15, hip_l2[:5]


def test_full_path_bs12_forward_matches_oracle(run_b12):
    _forward_report(run_b12)


def test_full_path_bs12_gradients_match_oracle(run_b12):
    """L2-relative error of 15 gradient tensors spread over join / GPT4 / layer4 / GPT1 / layer1 / stems against ONE
    fp32 oracle backward.  Bar: 2e-2 per tensor - two correct fp32 backward passes of this network differ by ~1e-3
    (median) of a tensor's largest entry because ReLU / max-pool decisions flip on rounding (fp64-calibrated below);
    a wrong scale, a missing term or a wrong mask offset gives O(1)."""
    _probe_gradients(run_b12, 2e-2)


def test_full_path_bs12_all_gradients_vs_fp64_yardstick(run_b12_f64):
    _all_gradients_vs_fp64(run_b12_f64)


def test_timed_configuration_bs12_dropout_forward_matches_oracle(run_b12_drop):
    """The configuration bench.py times: bs 12, n_layer 8, embd / attn / resid dropout 0.1.  Every dropout site of the
    forward (4 stages x (1 embd + 8 x (attn + 2 resid)) = 100 sites, ~1.4 G mask elements with per-site counter offsets
    off_e / off_a / off_p / off_m) must reproduce the oracle run on the CPU-rebuilt masks: stage taps, fused features,
    logits and loss to the 1e-3 bar."""
    _forward_report(run_b12_drop)


def test_timed_configuration_bs12_dropout_gradients(run_b12_drop):
    """... and its backward: the kernels regenerate every mask from offsets handed through the fused kernels
    (layernorm_bwd's dx_drop, avgpool_tokens_bwd's embd mask, the key-split attention backward with its dS / dropped-P
    hand-over at 1536 workgroups).  The same 15 probes as the dropout-free run; all tensors against fp64 below."""
    _probe_gradients(run_b12_drop, 2e-2)


def test_timed_configuration_bs12_dropout_all_gradients_vs_fp64_yardstick(run_b12_drop_f64):
    _all_gradients_vs_fp64(run_b12_drop_f64)


def test_timed_configuration_bs12_dropout_in_the_fp32_grade_split_mode(dev, run_b12_drop_f64):
    """`other_modes.f32x6` of the bench line (three-way bf16 split, six products: fp32-grade, faster than the exact fp32 MFMA on
    this part) at the configuration the bench times: the same weights, inputs, seed and salt as run_b12_drop, so the HIP path
    draws the very masks the cached oracle runs used.  Logits and loss against the fp32 oracle at the exact path's 1e-3 bar,
    ALL gradient tensors against the fp64 run with the fp32 oracle as yardstick - the bars of the exact path, unchanged."""
    from deepsense6g_tii_amd import ops
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    st = run_b12_drop_f64
    imgs, lids, rads, gps, target = st["inputs"]
    ops.set_compute_mode("f32x6")
    try:
        model = TransFuser(GlobalConfig(), dev)
        model.load_state_dict(st["sd"], strict=True)
        model.train()
        loss, logits = model.train_step_loss(imgs, lids, rads, gps, target)
        torch.cuda.synchronize()
    finally:
        ops.set_compute_mode("f32")
    assert (model._seed, model._salt_host) == (st["book"].seed, st["book"].salt) and model._drop_counter == st["sites_counted"]
    e_logits = rel(logits, st["ologits"])
    print(f"f32x6 at bs 12 with dropout: logits {e_logits:.2e} (exact path: {rel(st['logits'], st['ologits']):.2e}), "
          f"loss {float(loss):.6f} vs {st['oloss']:.6f}")
    assert e_logits < TOL and abs(float(loss) - st["oloss"]) < TOL * abs(st["oloss"])
    grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
    del model
    torch.cuda.empty_cache()
    _all_gradients_vs_fp64(dict(grads=grads, ograds=st["ograds"], grads64=st["grads64"]))


# ---------------------------------------------------------------------------------------------------------------------
def test_config1_image_only_zeroed_modalities(dev):
    """BASELINE configs[1]: LiDAR / radar inputs zeroed, bs = 12.  The all-zero stems give exactly zero-variance
    BatchNorm channels (invstd = 1/sqrt(eps)) in the forward AND the backward pass.  Exact fp32 mode against the oracle
    (stage taps, logits, loss, BN running statistics of the zero trunks), then bf16 mode: finite loss / gradients, logits
    inside the builder-declared bf16 bar (3e-2, tests/test_bf16_gpu.py)."""
    from deepsense6g_tii_amd import ops
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    kw = dict(embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    model, rcfg, sd = _build(dev, kw, seed=13)
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 12, seed=113)
    lids = [torch.zeros_like(t) for t in lids]
    rads = [torch.zeros_like(t) for t in rads]
    torch.set_num_threads(_threads())
    sdo = {k: v.clone() for k, v in sd.items()}
    ocap = {}
    with torch.no_grad():
        ologits = fr.transfuser_forward(sdo, imgs, lids, rads, gps, rcfg, fr.Ctx(training=True, capture=ocap))
        oloss = float(tr.sigmoid_focal_loss(ologits, target))
    model.train()
    cap = {}
    model._capture = cap
    loss, logits = model.train_step_loss(imgs, lids, rads, gps, target)
    model._capture = None
    torch.cuda.synchronize()
    for m in (1, 2):  # the two zero-input trunks, stage by stage
        for name in ("stem", "layer1", "layer2", "layer3", "layer4"):
            assert rel(nchw(cap[name][m]), ocap[name][m]) < TOL, (name, m)
    assert rel(logits, ologits) < TOL
    assert abs(float(loss) - oloss) < TOL * abs(oloss)
    bufs = dict(model.named_buffers())
    for trunk in ("encoder.lidar_encoder._model.", "encoder.radar_encoder._model."):
        # stem BN of an all-zero conv output: batch mean 0, batch var 0 -> running = 0.9 * init (+ 0.1 * 0)
        for stat in ("bn1.running_mean", "bn1.running_var", "layer1.0.bn1.running_mean", "layer1.0.bn1.running_var",
                     "layer4.1.bn2.running_var"):
            a, b = bufs[trunk + stat].cpu(), sdo[trunk + stat]
            assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item() + 1e-7, (trunk + stat)
    for n, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), n
    # zero-variance channels in the backward: the stem conv of a zero input has an exactly-zero weight gradient
    assert dict(model.named_parameters())["encoder.lidar_encoder._model.conv1.weight"].grad.abs().max().item() == 0.0
    # ---- the configuration as BASELINE names it: bf16 matrix-core mode ----
    model.zero_grad(set_to_none=True)
    ops.set_compute_mode("bf16")
    try:
        loss_b, logits_b = model.train_step_loss(imgs, lids, rads, gps, target)
        torch.cuda.synchronize()
    finally:
        ops.set_compute_mode("f32")
    assert torch.isfinite(loss_b).all() and torch.isfinite(logits_b).all()
    for n, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), n
    # (BN running statistics moved once more, so compare against the first step's logits with the bf16 bar)
    assert rel(logits_b, logits) < 3e-2, rel(logits_b, logits)


def test_config4_single_gpu_slice_bs32_ema_bf16_cosine(dev):
    """The per-GPU step of BASELINE configs[4]: bs = 32, EMA 0.999 fused into the optimizer kernel, bf16 matrix-core
    mode, lr from CyclicCosineDecayLR.  One step: finite loss, AdamW equals oracle.adamw_step on the step's own
    gradients at the scheduled lr, EMA shadow equals oracle.ema_update(old, new)."""
    from deepsense6g_tii_amd import ops
    from deepsense6g_tii_amd.synthetic import make_batch
    from deepsense6g_tii_amd.train import EMA, CyclicCosineDecayLR, FusedAdamW, train_iteration
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from oracle import train_ref as tr
    cfg = GlobalConfig()
    torch.manual_seed(100)
    model = TransFuser(cfg, dev)
    model.train()
    opt = FusedAdamW(model, lr=1e-4, ema_decay=0.999)
    ema = EMA(model, 0.999, opt)
    ema.register()
    sched = CyclicCosineDecayLR(opt)
    sched.step()  # epoch 1 of the warm-up
    lr = opt.param_groups[0]["lr"]
    assert abs(lr - tr.cyclic_cosine_lr(1, 1e-4)) < 1e-18
    batch = make_batch(32, cfg.seq_len, cfg.n_views, cfg.add_velocity, seed=104, device=dev)[:5]
    p0 = model.flat_parameters()[0].clone()
    ops.set_compute_mode("bf16")
    try:
        loss, logits = train_iteration(model, opt, batch, ema)
        torch.cuda.synchronize()
    finally:
        ops.set_compute_mode("f32")
    assert torch.isfinite(loss).all() and logits.shape == (32, 64)
    p1, g = model.flat_parameters()
    assert torch.isfinite(g).all()
    ref = p0.cpu().clone()
    m, v = torch.zeros_like(ref), torch.zeros_like(ref)
    tr.adamw_step(ref, g.cpu(), m, v, 1, lr)
    assert (p1.cpu() - ref).abs().max().item() < 1e-7 + 1e-6 * ref.abs().max().item()
    sh = tr.ema_update(p0.cpu(), p1.cpu(), 0.999)
    assert (opt.shadow.cpu() - sh).abs().max().item() < 1e-7


# ---------------------------------------------------------------------------------------------------------------------
def test_dropout_masks_are_plumbed_consistently_at_model_level(dev):
    """Train mode with the reference's dropout 0.1 on all three sites.  The masks are a pure function of (seed, counter):
    this test re-derives every site's counter offset from the documented order (the step's salt, then per stage:
    embd_drop on (B, T, C); per block: attn_drop on (B, nh, T, T), resid_drop after proj, resid_drop after the MLP; each
    advanced to the next multiple of 1024), rebuilds the masks on the CPU and runs the ORACLE on them.  Forward logits must match to 1e-3 and
    the gradients (whose backward kernels regenerate the masks from offsets handed over through fused kernels:
    layernorm_bwd's dx_drop, avgpool_tokens_bwd's embd mask, the attention backward) like in the dropout-free test."""
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    kw = dict(n_layer=2)          # dropout 0.1 / 0.1 / 0.1 from the config defaults
    model, rcfg, sd = _build(dev, kw, seed=17)
    B = 2
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, B, seed=117)
    model.train()
    loss, logits = model.train_step_loss(imgs, lids, rads, gps, target)
    torch.cuda.synchronize()
    # effective counter offset of a site = the step's salt (device-resident, advanced once per training forward; the host
    # mirrors it) + the site's offset inside the forward, which restarts at 0 every step
    seed, salt = model._seed, model._salt_host
    assert salt == int(model._salt.item()) == model.SALT_STRIDE
    counter = [salt]

    def mask_fn(shape, p):
        n = int(np.prod(shape))
        off = counter[0]
        counter[0] += (n + 1023) // 1024 * 1024
        return _keep_mask(seed, off, shape, p)

    torch.set_num_threads(_threads())
    sdo = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not fr.is_buffer(k)) else v.clone())
           for k, v in sd.items()}
    ologits = fr.transfuser_forward(sdo, imgs, lids, rads, gps, rcfg, fr.Ctx(training=True, dropout=True, mask_fn=mask_fn))
    assert counter[0] - salt == model._drop_counter      # the oracle visited exactly the sites the HIP walk counted
    oloss = tr.sigmoid_focal_loss(ologits, target)
    oloss.backward()
    assert rel(logits, ologits.detach()) < TOL
    assert abs(float(loss) - float(oloss)) < TOL * abs(float(oloss))
    worst = []
    for n in ("join.0.weight", "encoder.transformer4.blocks.1.mlp.2.weight", "encoder.transformer4.blocks.0.attn.value.weight",
              "encoder.transformer3.blocks.1.attn.proj.weight", "encoder.transformer2.pos_emb",
              "encoder.transformer1.blocks.0.mlp.0.weight", "encoder.transformer1.blocks.0.attn.query.weight",
              "encoder.image_encoder.features.layer3.0.conv1.weight", "encoder.radar_encoder._model.layer1.1.conv2.weight"):
        worst.append((l2rel(dict(model.named_parameters())[n].grad, sdo[n].grad), n))
    print("\n".join(f"{e:.3e} {n}" for e, n in worst))
    # a wrong mask offset in any backward kernel decorrelates ~19 % of that site's gradient entries: O(0.3) here
    assert max(worst)[0] < 5e-2, max(worst)


def test_grad_norm_clip_matches_torch(dev):
    """fused global-norm clip of the 30->5 training step (train2_seq_30to5.py:120): norm + coefficient from one reduction
    kernel, applied inside the AdamW kernel - against torch.nn.utils.clip_grad_norm_ + oracle AdamW."""
    from deepsense6g_tii_amd.train import FusedAdamW
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    kw = dict(n_layer=1, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    model, rcfg, sd = _build(dev, kw, seed=19)
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, 1, seed=119)
    model.train()
    for max_norm in (1e-3, 1e6):          # clipping active / inactive
        opt = FusedAdamW(model, lr=1e-3, max_grad_norm=max_norm)
        opt.zero_grad()
        model.train_step_loss(imgs, lids, rads, gps, target)
        p0 = model.flat_parameters()[0].cpu().clone()
        gflat = model.flat_parameters()[1].cpu().clone()
        plist = [torch.nn.Parameter(torch.zeros_like(p, device="cpu")) for p in model.parameters()]
        for q, p in zip(plist, model.parameters()):
            q.grad = p.grad.detach().cpu().clone()
        total = torch.nn.utils.clip_grad_norm_(plist, max_norm)
        opt.step()
        torch.cuda.synchronize()
        assert abs(opt.last_grad_norm() - float(total)) < 1e-5 * float(total)
        coef = min(1.0, max_norm / (float(total) + 1e-6))
        ref = p0.clone()
        m, v = torch.zeros_like(ref), torch.zeros_like(ref)
        tr.adamw_step(ref, gflat * coef, m, v, 1, 1e-3)
        assert (model.flat_parameters()[0].cpu() - ref).abs().max().item() < 1e-7 + 2e-6 * ref.abs().max().item()
        assert torch.equal(model.flat_parameters()[1].cpu(), gflat)      # the arena itself is not rewritten
