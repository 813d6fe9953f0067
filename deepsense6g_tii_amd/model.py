"""Host side of the fusion hot path: a drop-in for the reference's ``TransFuser`` boundary.

Mirrors /root/reference/model2_seq.py ``TransFuser(config, device, pretrain_weight=False)`` and
``forward(image_list, lidar_list, radar_list, gps, rebuild_modality_feat_list=None) -> (B,64)``
(:850-894) with the GPT ``Encoder`` (:406-597) wired in.  Parameter names / shapes equal the
reference state-dict, so checkpoints interchange.

What runs where: Python only walks the layer list and hands device pointers to libds6g.so
(include/ds6g.h).  torch is used for HBM allocations, the current HIP stream and autograd glue
(one autograd.Function around the whole path; its backward is the hand-written reverse walk
below, which writes parameter gradients straight into a flat gradient arena).

Memory layout (MI355X-first):
  * all parameters are views into ONE flat fp32 arena, all gradients into a second one, so the
    optimizer is a single streaming kernel and data-parallel all-reduce runs on contiguous chunks
    without packing copies;
  * conv weights are stored OHWI (torch channels_last views of OIHW-shaped parameters);
  * activations are NHWC, so the 8x8 pooled maps are rows of the (B, 962, C) token buffer.
Kernels read parameter pointers at call time (the reference's EMA re-points ``param.data``,
train2_seq.py:326-333).
"""
from __future__ import annotations

import math
import os

import torch
from torch import nn

from . import ops
from ._lib import lib

F32 = torch.float32
STAGE_WIDTH = (64, 128, 256, 512)
RESNET_LAYERS = {"resnet34": (3, 4, 6, 3), "resnet18": (2, 2, 2, 2)}


class GlobalConfig:
    """Same attribute names / defaults as /root/reference/config_seq.py:3-45."""
    seq_len = 5
    pred_len = 4
    data_root = "./Dataset"
    n_views = 1
    input_resolution = 256
    scale = 1
    crop = 256
    lr = 1e-4
    FFM = 1
    TFM = 1
    modality_missing = None
    modality_missing_type = "zerolike"
    vert_anchors = 8
    horz_anchors = 8
    anchors = vert_anchors * horz_anchors
    n_embd = 512
    block_exp = 4
    n_layer = 8
    n_head = 4
    n_scale = 4
    embd_pdrop = 0.1
    resid_pdrop = 0.1
    attn_pdrop = 0.1
    add_velocity = 1

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)


# ------------------------------------------------------------------------------------------------
# parameter containers: plain torch modules used ONLY to hold parameters/buffers under the
# reference's state-dict names; their forward() is never called.
# ------------------------------------------------------------------------------------------------
class _BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride


class _ResNetTrunk(nn.Module):
    def __init__(self, arch, in_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for li, (planes, n) in enumerate(zip(STAGE_WIDTH, RESNET_LAYERS[arch]), start=1):
            blocks = []
            for bi in range(n):
                blocks.append(_BasicBlock(inpl, planes, 2 if (li > 1 and bi == 0) else 1))
                inpl = planes
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Sequential()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class ImageCNN(nn.Module):  # model2_seq.py:12-34
    def __init__(self, c_dim=512, normalize=True):
        super().__init__()
        self.normalize = normalize
        self.features = _ResNetTrunk("resnet34", 3)


class LidarEncoder(nn.Module):  # model2_seq.py:48-72
    def __init__(self, num_classes=512, in_channels=2):
        super().__init__()
        self._model = _ResNetTrunk("resnet18", in_channels)


class _SelfAttention(nn.Module):  # model2_seq.py:74-91
    def __init__(self, n_embd, n_head):
        super().__init__()
        assert n_embd % n_head == 0
        self.key = nn.Linear(n_embd, n_embd)
        self.query = nn.Linear(n_embd, n_embd)
        self.value = nn.Linear(n_embd, n_embd)
        self.proj = nn.Linear(n_embd, n_embd)
        self.n_head = n_head


class _Block(nn.Module):  # model2_seq.py:113-126
    def __init__(self, n_embd, n_head, block_exp):
        super().__init__()
        self.ln1 = nn.LayerNorm(n_embd)
        self.ln2 = nn.LayerNorm(n_embd)
        self.attn = _SelfAttention(n_embd, n_head)
        self.mlp = nn.Sequential(nn.Linear(n_embd, block_exp * n_embd), nn.ReLU(True),
                                 nn.Linear(block_exp * n_embd, n_embd), nn.Dropout(0.0))


class GPT(nn.Module):  # model2_seq.py:175-214
    def __init__(self, n_embd, config):
        super().__init__()
        self.n_embd = n_embd
        n_tok = (config.n_views + 2) * config.seq_len * config.vert_anchors * config.horz_anchors + 2
        self.pos_emb = nn.Parameter(torch.zeros(1, n_tok, n_embd))
        self.blocks = nn.Sequential(*[_Block(n_embd, config.n_head, config.block_exp) for _ in range(config.n_layer)])
        self.ln_f = nn.LayerNorm(n_embd)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(mean=0.0, std=0.02)
                m.bias.data.zero_()


class Encoder(nn.Module):  # model2_seq.py:406-470
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.image_encoder = ImageCNN(512, normalize=True)
        self.lidar_encoder = LidarEncoder(512, in_channels=1)
        self.radar_encoder = LidarEncoder(512, in_channels=2 if config.add_velocity else 1)
        self.vel_emb1 = nn.Linear(2, 64)
        self.vel_emb2 = nn.Linear(64, 128)
        self.vel_emb3 = nn.Linear(128, 256)
        self.vel_emb4 = nn.Linear(256, 512)
        self.transformer1 = GPT(64, config)
        self.transformer2 = GPT(128, config)
        self.transformer3 = GPT(256, config)
        self.transformer4 = GPT(512, config)


# ------------------------------------------------------------------------------------------------
class _FusionFn(torch.autograd.Function):
    """Autograd glue: forward = kernel walk, backward = hand-written reverse walk that deposits
    parameter gradients in the gradient arena (no gradient is returned for the anchor)."""

    @staticmethod
    def forward(ctx, anchor, model, images, lidars, radars, gps):
        logits, tape = model._run_forward(images, lidars, radars, gps, record=True)
        ctx.model = model
        ctx.tape = tape
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        ctx.model._run_backward(ctx.tape, dlogits.contiguous())
        ctx.tape = None
        return None, None, None, None, None, None


class TransFuser(nn.Module):
    """Drop-in for model2_seq.TransFuser (GPT variant).  See module docstring."""
    _GRU_HEAD = False

    def __init__(self, config, device, pretrain_weight=False):
        super().__init__()
        self.device = torch.device(device)
        self.config = config
        self.pred_len = getattr(config, "pred_len", 4)
        self.encoder = Encoder(config)
        self.join = nn.Sequential(nn.Linear(512, 256), nn.ReLU(inplace=True), nn.Linear(256, 128),
                                  nn.ReLU(inplace=True), nn.Linear(128, 64))
        # the 30->5 variant (model2_seq_30to5.py:842-843) appends an autoregressive GRU head; TransFuser30to5 sets this
        self.gru_head = bool(getattr(config, "gru_head", False)) or self._GRU_HEAD
        if self.gru_head:
            self.decoder = nn.GRUCell(input_size=64, hidden_size=64)
            self.output = nn.Linear(64, 64)
        if pretrain_weight:
            self.load_pretrained_weight()
        self._seed = self._base_seed = 0x5DEECE66D
        self._seed_rank = 0
        self._drop_counter = 0   # counter offset of the next dropout site inside the CURRENT forward (restarts at 0 every forward)
        # device-resident dropout salt: advanced by SALT_STRIDE on the device at the start of every training forward and
        # added to every mask counter by the kernels at run time (ds6g_set_dropout_salt), so the launch arguments of a step
        # are the same every step (hipGraph-replayable) while the masks are fresh; _salt_host mirrors it on the host
        self._salt = None
        self._salt_host = 0
        self._ws_main = None
        self._ws_side = {}
        self._side_streams = None
        self.multi_stream = True  # run the three (independent) trunks on three HIP streams between fusion points
        self.overlap_wgrad = True  # GPT-stage weight gradients on a second stream, overlapping the dgrad / attention chain
        self._wg_map, self._wg_used, self._wg_keep = {}, {}, []
        self.overlap_wgrad_trunks = False  # measured: no gain on top of the three concurrent trunk streams
        # bf16 configuration: train-mode BatchNorm statistics come out of the conv's epilogue (ds6g_bf16_conv2d_fwd_bnstats)
        self.fuse_bn_stats16 = os.environ.get("DS6G_FUSE_BN_STATS16", "1") != "0"
        self._fold_now = False
        self._recording = False
        self.use_winograd = True  # 3x3 / stride-1 convs (forward and data gradient) as Winograd F(2x2, 3x3) in fp32 mode
        self.fold_bn_eval = True  # eval(): BatchNorm folded into the conv weights (no BN kernels at inference)
        self.fuse_qkv = True      # key|query|value projections as one GEMM when their parameters are contiguous (arena)
        # bf16 matrix-core mode ("bf16 forward/backward", BASELINE configs[1] / [4]): GEMM operands of the GPT stages are
        # STORED as bf16 (LayerNorm outputs, attention output, MLP hidden, their gradients, a bf16 shadow of the weights
        # refreshed once per forward; csrc/bgemm.hip) instead of fp32 tiles rounded on the way into the MFMA.  fp32 stay: the
        # master weights, the residual stream, LayerNorm / softmax statistics, every accumulator, loss, optimizer.
        self.bf16_storage = True
        self.bf16_stems = os.environ.get("DS6G_BF16_STEMS", "1") != "0"   # 7x7 stems on bf16 storage too (csrc/stem.hip)
        self._arena16 = None
        self._use16 = False
        self._anchor = None
        self._arena = None
        self._wfast = None     # parameter-pointer table of the walk in flight (see _run_forward)
        self._wtable = None
        if self.device.type == "cuda":
            lib()  # fail loudly now if the HIP library is missing
            self._build_arena()

    def load_pretrained_weight(self):  # model2_seq.py:875-878
        self.load_state_dict(torch.load("mamba_fusion.pth", weights_only=True))

    # ---------------------------------------------------------------- arenas --------------------
    @staticmethod
    def _milestone(name):
        """Backward completion order of a parameter's gradient: 0 = join (first ready) ... 9 = stems (last).
        The arenas are laid out in this order so that, while the backward walk runs, the finished
        gradients always form ONE contiguous, growing prefix of the gradient arena: data-parallel
        all-reduce buckets are plain slices of it (no packing copy) and can start while earlier
        layers are still being differentiated."""
        if name.startswith(("join.", "decoder.", "output.")):
            return 0
        for s in (4, 3, 2, 1):
            if f"transformer{s}." in name or f"vel_emb{s}." in name:
                return 1 + 2 * (4 - s)
            if f".layer{s}." in name:
                return 2 + 2 * (4 - s)
        return 9

    @staticmethod
    def _group_qkv(named):
        """Lay the key / query / value projections of every SelfAttention (model2_seq.py:83-85) out as ONE
        [3C, C] weight block followed by ONE [3C] bias block, so the three projections run as a single GEMM
        (and their gradients as a single wgrad / dgrad) straight on the arena."""
        order = {"key.weight": 0, "query.weight": 1, "value.weight": 2, "key.bias": 3, "query.bias": 4, "value.bias": 5}
        out, i = [], 0
        while i < len(named):
            name = named[i][0]
            if name.endswith(".attn.key.weight"):
                grp = named[i:i + 6]
                pref = name[: -len("key.weight")]
                assert all(n.startswith(pref) and n[len(pref):] in order for n, _ in grp), [n for n, _ in grp]
                out.extend(sorted(grp, key=lambda kv: order[kv[0][len(pref):]]))
                i += 6
            else:
                out.append(named[i])
                i += 1
        return out

    def _qkv_fused(self, at, grads=False):
        """(weight ptr, bias ptr) of the fused key|query|value block when the three projections are contiguous in
        memory right now (arena or EMA-shadow layout; param.data may have been re-pointed), else None."""
        C = at.key.weight.shape[0]
        if grads:
            w = [self._g(m.weight) for m in (at.key, at.query, at.value)]
            b = [self._g(m.bias) for m in (at.key, at.query, at.value)]
            if len({f for _, f in w + b}) != 1:
                return None
            wp, bp = [x for x, _ in w], [x for x, _ in b]
        else:
            wp = [self._w(m.weight) for m in (at.key, at.query, at.value)]
            bp = [self._w(m.bias) for m in (at.key, at.query, at.value)]
        if wp[1] == wp[0] + 4 * C * C and wp[2] == wp[0] + 8 * C * C and bp[1] == bp[0] + 4 * C and bp[2] == bp[0] + 8 * C:
            return wp[0], bp[0]
        return None

    def arena_layout(self):
        """-> (ordered [(name, param)], {name: (offset, numel)}, {milestone: end offset}, total elements) of the flat
        parameter / gradient arenas: parameters sorted by backward-completion milestone, key|query|value grouped, every
        segment padded to 4 elements.  Pure host arithmetic (also valid for a model constructed on "cpu", where no arena
        is allocated): the data-parallel bucket table is a function of this layout alone."""
        named = sorted(self.named_parameters(), key=lambda kv: self._milestone(kv[0]))  # stable
        named = self._group_qkv(named)
        pslice, milestone_end, off = {}, {}, 0
        for name, p in named:
            pslice[name] = (off, p.numel())
            off += (p.numel() + 3) // 4 * 4
            milestone_end[self._milestone(name)] = off
        return named, pslice, milestone_end, off

    def _build_arena(self):
        dev = self.device
        named, _, _, total = self.arena_layout()
        self._arena = torch.zeros(total, dtype=F32, device=dev)
        self._garena = torch.zeros(total, dtype=F32, device=dev)
        self._gview = {}
        self._pslice = {}
        self._milestone_end = {}
        self.grad_ready_hook = None  # callable(milestone, lo, hi): grads garena[lo:hi] are final
        off = 0
        for name, p in named:
            n = p.numel()
            seg = self._arena[off:off + n]
            gseg = self._garena[off:off + n]
            if p.dim() == 4:  # conv OIHW parameter stored OHWI (channels_last)
                O, I, R, S = p.shape
                view = seg.view(O, R, S, I).permute(0, 3, 1, 2)
                gview = gseg.view(O, R, S, I).permute(0, 3, 1, 2)
            else:
                view = seg.view(p.shape)
                gview = gseg.view(p.shape)
            view.copy_(p.data)
            p.data = view
            self._gview[name] = gview
            self._pslice[name] = (off, n)
            off += (n + 3) // 4 * 4
            self._milestone_end[self._milestone(name)] = off
        self._arena_used = off
        for _, b in self.named_buffers():
            b.data = b.data.to(dev)
        nb = [b for n_, b in self.named_buffers() if n_.endswith("num_batches_tracked")]
        self._nbt = torch.zeros(len(nb), dtype=torch.long, device=dev)
        for i, b in enumerate(nb):
            b.data = self._nbt[i]
        self._ws_main = ops.Workspace(dev, 1 << 30)
        self._anchor = torch.zeros(1, dtype=F32, device=dev, requires_grad=True)
        self._pname = {id(p): n for n, p in named}
        # the hot path walks the parameter list several times per step (fresh / accumulate modes, arena check, zero_grad):
        # nn.Module.named_parameters() re-traverses the module tree each time (~3 ms of host per walk), the set is fixed
        self._plist = list(self.named_parameters())
        a0 = self._arena.data_ptr()
        self._wtable = {id(p): a0 + 4 * self._pslice[n][0] for n, p in self._plist}

    @property
    def _ws(self):
        """scratch of the stream the caller is launching on (a scratch buffer is only safe within one stream)"""
        if self._ws_side:
            ws = self._ws_side.get(ops._stream())
            if ws is not None:
                return ws
        return self._ws_main

    def _attn_ws(self, B, T, nh, C):
        """scratch for the attention backward, grown (main stream only) to what lets the dK/dV kernel hand its dS / P
        tiles to the dQ / dV kernels instead of every kernel recomputing the scores (ds6g_attention_workspace_bytes)"""
        ws = self._ws
        if ws is self._ws_main:
            need = int(lib().attention_workspace_bytes(B, T, nh, C // nh, C))
            if need > ws.nbytes:
                self._ws_main = ws = ops.Workspace(self.device, need)
        return ws

    def _fork(self):
        """-> the three trunk streams, each ordered after everything enqueued so far on the current stream.
        Discipline that keeps the caching allocator safe without record_stream(): between _fork() and _join() the
        calling stream launches nothing, every tensor a trunk stream allocates is used on that stream only until
        the join, and tensors crossing the boundary are kept alive by the tape."""
        if self._side_streams is None:
            # (a higher stream priority for the camera trunk, which carries twice the work of the other two, was measured:
            # 170 -> 124 samples/s - priority queues serialise against the default-priority streams on this runtime)
            self._side_streams = [torch.cuda.Stream(self.device) for _ in range(3)]
            for st in self._side_streams:
                self._ws_side[st.cuda_stream] = ops.Workspace(self.device, 256 << 20)
        cur = ops.current_stream_obj()
        for st in self._side_streams:
            st.wait_stream(cur)
        return self._side_streams

    def _join(self):
        cur = ops.current_stream_obj()
        for st in self._side_streams:
            cur.wait_stream(st)

    def _trunk_ctx(self, streams, m):
        import contextlib
        return ops.on_stream(streams[m]) if streams is not None else contextlib.nullcontext()

    def _apply(self, fn, recurse=True):
        # .to(same device) keeps the arena views; a real move would silently detach parameters from it
        out = super()._apply(fn, recurse)
        if self._arena is not None and not self.params_in_arena():
            raise RuntimeError("TransFuser owns its device memory; construct it with the target device")
        return out

    def flat_parameters(self):
        """(param_arena, grad_arena) - contiguous fp32 buffers behind all parameters / gradients."""
        return self._arena[: self._arena_used], self._garena[: self._arena_used]

    def params_in_arena(self):
        a0 = self._arena.data_ptr()
        for name, p in self._plist:
            off, _ = self._pslice[name]
            if p.data_ptr() != a0 + 4 * off:
                return False
        return True

    # ---------------------------------------------------------------- pointers ------------------
    def _w(self, p):
        """device pointer of a parameter as the kernels expect it (conv: OHWI)."""
        fast = self._wfast
        if fast is not None:      # walk in flight with every parameter in the arena (checked once at its start)
            return fast[id(p)]
        if p.dim() == 4 and not p.data.is_contiguous(memory_format=torch.channels_last):
            p.data = p.data.contiguous(memory_format=torch.channels_last)
        elif p.dim() != 4 and not p.data.is_contiguous():
            p.data = p.data.contiguous()
        return p.data_ptr()

    def _g(self, p):
        """(grad pointer, accumulate flag) of parameter p for the backward walk in flight."""
        return self._gmode[id(p)]

    def _w16(self, p):
        """device pointer of parameter p inside the bf16 shadow of the parameter arena (same layout, 2-byte elements)"""
        return self._arena16.data_ptr() + 2 * self._pslice[self._pname[id(p)]][0]

    def _refresh_shadow16(self):
        """bf16 shadow of the fp32 master weights, refreshed once per forward of the bf16-storage path: one streaming cast
        over the arena (0.47 GB of traffic, ~0.1 ms: 0.3 % of a step), so it can never be stale whatever touched the
        parameters (optimizer, load_state_dict, in-place edits).  Off when the parameters were re-pointed away from the
        arena (EMA shadow applied): the fp32-storage kernels then run, reading the live pointers."""
        self._use16 = bool(self.bf16_storage and self._arena is not None and lib().get_compute_mode() == 1
                           and self.fuse_qkv and self.params_in_arena())
        if self._use16:
            if self._arena16 is None:
                self._arena16 = torch.empty(self._arena.numel(), dtype=torch.bfloat16, device=self.device)
            ops.cast_bf16(self._arena, out=self._arena16)

    def _begin_backward(self):
        # .grad is None -> write fresh and attach the arena view; .grad is our view -> accumulate in
        # place (torch semantics when zero_grad was not called); foreign tensor -> write, then add.
        self._gmode, self._fresh, self._foreign = {}, [], []
        plist = self._plist if self._arena is not None else list(self.named_parameters())
        if self.grad_ready_hook is not None:
            self._dp_check_fresh_grads()
            begin = getattr(self.grad_ready_hook, "__self__", None)
            if begin is not None and hasattr(begin, "begin"):
                begin.begin()
        for name, p in plist:
            gv = self._gview[name]
            if p.grad is None:
                self._fresh.append((p, gv))
                self._gmode[id(p)] = (gv.data_ptr(), 0)
            elif p.grad.data_ptr() == gv.data_ptr():
                self._gmode[id(p)] = (gv.data_ptr(), 1)
            else:
                self._foreign.append((p, gv))
                self._gmode[id(p)] = (gv.data_ptr(), 0)

    def _milestone_done(self, k):
        if self.grad_ready_hook is not None:
            lo = self._milestone_end.get(k - 1, 0) if k > 0 else 0
            self.grad_ready_hook(k, lo, self._milestone_end[k])

    def _end_backward(self):
        self._wfast = None
        owner = getattr(self.grad_ready_hook, "__self__", None)
        if owner is not None and hasattr(owner, "finish"):
            owner.finish()  # flush the tail bucket; the calling stream waits for every outstanding all-reduce
        for p, gv in self._fresh:
            p.grad = gv
        for p, gv in self._foreign:
            p.grad.add_(gv)
        self._gmode, self._fresh, self._foreign = {}, [], []

    def set_dropout_seed(self, seed, rank=0):
        """Seed of the counter-based dropout masks.  Data-parallel ranks must draw independent masks (the reference's
        DataParallel replicas each use their own device RNG): the rank is mixed in with a splitmix64 finalizer.  The
        UNMIXED base seed and the rank are kept: a checkpoint carries the base seed, and loading it re-applies this
        rank's mix (rng_state / set_rng_state) - a rank-0 checkpoint never collapses the ranks onto one mask stream."""
        self._base_seed, self._seed_rank = int(seed) & 0xFFFFFFFFFFFFFFFF, int(rank)
        z = (int(seed) + 0x9E3779B97F4A7C15 * (int(rank) + 1)) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        self._seed = (z ^ (z >> 31)) if rank else self._base_seed
        return self._seed

    SALT_STRIDE = 1 << 40   # counter space of one forward (bs=32: 4e9 attention-mask elements per step)

    def rng_state(self):
        """(base seed, salt) of the dropout stream - saved with the optimizer state so a resumed run continues the mask
        sequence instead of replaying it from step 0.  `seed` is the UNMIXED base seed (identical on every rank)."""
        return dict(seed=int(self._base_seed), counter=int(self._salt_host))

    def set_rng_state(self, st):
        """restores the mask counter and the base seed; the seed is re-mixed with THIS model's rank (set by
        set_dropout_seed / dist.attach), whichever of attach() and load_state_dict() ran first"""
        self.set_dropout_seed(int(st["seed"]), self._seed_rank)
        self._salt_host = int(st["counter"])
        if self._salt is not None:
            self._salt.fill_(self._salt_host)

    def _advance_salt(self):
        """start of a training forward: new mask epoch (device add: capturable), counters restart at 0"""
        if self._salt is None:
            self._salt = torch.full((1,), self._salt_host, dtype=torch.int64, device=self.device)
        self._salt.add_(self.SALT_STRIDE)
        self._salt_host += self.SALT_STRIDE
        self._drop_counter = 0
        # this forward's own copy: its backward regenerates the masks from it even if another forward ran in between
        self._salt_cur = self._salt.clone()

    def _note_replayed_step(self):
        """a captured training step was replayed: the device advanced the salt, the host mirror follows"""
        self._salt_host += self.SALT_STRIDE

    def _next_drop(self, numel):
        off = self._drop_counter
        self._drop_counter += (int(numel) + 1023) // 1024 * 1024
        return off

    # ---------------------------------------------------------------- public --------------------
    def _inputs(self, image_list, lidar_list, radar_list, gps):
        """reference-style lists of NCHW frames (cast to fp32 on the device as Engine.train does, train2_seq.py:
        111-116) - or one data.PackedInputs whose tensors are already in the stem's NHWC x4 layout."""
        from .data import PackedInputs
        cfg = self.config
        if isinstance(image_list, PackedInputs):
            pk = image_list
            assert lidar_list is None and radar_list is None and gps is None, "PackedInputs carries every modality"
            assert pk.seq_len == cfg.seq_len, (pk.seq_len, cfg.seq_len)
            cfg.n_views = 1
            return pk.images, pk.lidars, pk.radars, pk.gps.to(self.device, F32).contiguous()
        cfg.n_views = len(image_list) // cfg.seq_len  # side effect kept from model2_seq.py:489
        images = [t.to(self.device, F32).contiguous() for t in image_list]
        lidars = [t.to(self.device, F32).contiguous() for t in lidar_list]
        radars = [t.to(self.device, F32).contiguous() for t in radar_list]
        return images, lidars, radars, gps.to(self.device, F32).contiguous()

    def forward(self, image_list, lidar_list=None, radar_list=None, gps=None, rebuild_modality_feat_list=None):
        if self.device.type != "cuda":
            raise RuntimeError("deepsense6g_tii_amd.TransFuser runs on MI355X HIP kernels only (no CPU path)")
        images, lidars, radars, gps = self._inputs(image_list, lidar_list, radar_list, gps)
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            return _FusionFn.apply(self._anchor, self, images, lidars, radars, gps)
        logits, _ = self._run_forward(images, lidars, radars, gps, record=False)
        return logits

    def capture_inference(self, image_list, lidar_list, radar_list, gps):
        """Serving path: captures the eval-mode forward for inputs of these shapes into ONE HIP graph and returns
        ``run(image_list, lidar_list, radar_list, gps) -> logits``, which copies the new inputs into the graph's static
        buffers and replays it.  A single-sample forward is launch-bound from Python (~600 kernel launches for ~3 ms of
        GPU work); replayed from the graph the host cost is one call.  The graph reads parameter memory at replay time
        (weights may keep training / be swapped by EMA between calls as long as they stay in the same storage); the
        returned tensor is overwritten by the next replay."""
        if self.training:
            raise RuntimeError("capture_inference() needs model.eval()")
        to_dev = lambda seq: [t.to(self.device, F32).contiguous().clone() for t in seq]  # noqa: E731
        st_img, st_lid, st_rad = to_dev(image_list), to_dev(lidar_list), to_dev(radar_list)
        st_gps = gps.to(self.device, F32).contiguous().clone()
        cur = torch.cuda.current_stream()
        warm = torch.cuda.Stream(self.device)
        warm.wait_stream(cur)
        with torch.cuda.stream(warm), torch.no_grad():  # lazy one-time work (side streams, scratch) outside the capture
            for _ in range(2):
                self.forward(st_img, st_lid, st_rad, st_gps)
        cur.wait_stream(warm)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            out = self.forward(st_img, st_lid, st_rad, st_gps)

        def run(image_list, lidar_list, radar_list, gps):
            for dst, src in zip(st_img + st_lid + st_rad + [st_gps], list(image_list) + list(lidar_list) + list(radar_list) + [gps]):
                dst.copy_(src, non_blocking=True)
            graph.replay()
            return out
        run.graph = graph
        return run

    def train_step_loss(self, image_list, lidar_list, radar_list, gps, target, alpha=0.25, gamma=2.0):
        """Fused forward -> sigmoid focal loss -> backward without autograd (the harness path).
        Returns (loss tensor [1], logits)."""
        images, lidars, radars, gps = self._inputs(image_list, lidar_list, radar_list, gps)
        if target.dim() == 1:  # class-index target (temp_coef = 0 path, train2_seq.py:124 -> FocalLoss :297-298)
            target = torch.nn.functional.one_hot(target.long(), num_classes=64)
        target = target.to(self.device, F32).contiguous()
        logits, tape = self._run_forward(images, lidars, radars, gps, record=True)
        if tuple(target.shape) != tuple(logits.shape):  # the kernel reads logits.numel() floats from both buffers
            raise ValueError(f"focal-loss target shape {tuple(target.shape)} != logits shape {tuple(logits.shape)}")
        loss = torch.empty(1, dtype=F32, device=self.device)
        dlogits = torch.empty_like(logits)
        lib().focal_loss(logits.data_ptr(), target.data_ptr(), loss.data_ptr(), dlogits.data_ptr(), logits.numel(),
                         alpha, gamma, 1.0, ops._stream())
        self._run_backward(tape, dlogits)
        return loss, logits

    # ================================================================ forward walk ==============
    def _trunks(self):
        e = self.encoder
        return ((e.image_encoder.features, "resnet34", 3, True),
                (e.lidar_encoder._model, "resnet18", 1, False),
                (e.radar_encoder._model, "resnet18", 2 if self.config.add_velocity else 1, False))

    def _bn_fwd(self, bn, x, relu, residual, train):
        C = x.shape[-1]
        M = x.numel() // C
        stats = torch.empty(2, C, dtype=F32, device=x.device)
        mean, invstd = stats[0], stats[1]
        if train:
            ops.bn_stats(M, C, x, mean, invstd, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self._ws,
                         bn.eps, bn.momentum)
        else:
            ops.bn_eval_prepare(bn.running_mean.data_ptr(), bn.running_var.data_ptr(), C, mean, invstd, bn.eps)
        y = ops.bn_apply(x, mean, invstd, self._w(bn.weight), self._w(bn.bias), relu, residual)
        return y, (mean, invstd)

    def _stem_fwd(self, trunk, cin, normalize, frames, train):
        L = lib()
        st = ops._stream()
        if (self._use16 and self.bf16_stems and not self._fold_now and not torch.is_tensor(frames)
                and ops.bf16_stem_ok(*frames[0].shape[2:])):
            # bf16 configuration: the stem too on bf16 storage (csrc/stem.hip) - packed input, conv output and its gradient
            # are bf16; the conv's epilogue delivers the BatchNorm statistics
            B, S = frames[0].shape[0], len(frames)
            H, W = frames[0].shape[2:]
            x16 = torch.empty((B * S, H, W, 4), dtype=torch.bfloat16, device=self.device)
            for t, f in enumerate(frames):
                assert f.shape == (B, cin, H, W), (f.shape, (B, cin, H, W))
                L.pack_input_bf16(f.data_ptr(), x16.data_ptr(), B, cin, H, W, S, t, int(normalize), st)
            bn = trunk.bn1
            stats = torch.empty(2, 64, dtype=F32, device=self.device)
            if train:
                c1 = ops.bf16_stem_fwd(x16, self._w(trunk.conv1.weight), cin, self._ws, (stats[0], stats[1]),
                                       bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.eps, bn.momentum)
            else:
                c1 = ops.bf16_stem_fwd(x16, self._w(trunk.conv1.weight), cin, self._ws)
                ops.bn_eval_prepare(bn.running_mean.data_ptr(), bn.running_var.data_ptr(), 64, stats[0], stats[1], bn.eps)
            p1, idx = ops.bf16_stem_bn_relu_maxpool(c1, stats[0], stats[1], self._w(bn.weight), self._w(bn.bias))
            return p1, (x16, c1, None, (stats[0], stats[1]), idx, cin)
        if torch.is_tensor(frames):  # data.PackedInputs: already NHWC x4, normalised
            x = frames
            assert x.dim() == 4 and x.shape[3] == 4 and x.dtype == F32 and x.is_contiguous() and x.device == self.device
        else:
            B = frames[0].shape[0]
            S = len(frames)
            H, W = frames[0].shape[2:]
            x = torch.empty((B * S, H, W, 4), dtype=F32, device=self.device)
            for t, f in enumerate(frames):
                assert f.shape == (B, cin, H, W), (f.shape, (B, cin, H, W))
                L.pack_input(f.data_ptr(), x.data_ptr(), B, cin, H, W, 4, S, t, int(normalize), st)
        if self._fold_now:
            wf, bf = ops.bn_fold(self._w(trunk.conv1.weight), trunk.bn1, 64, 49, cin, 4)  # folds and pads to 4 channels
            c1, st1 = None, None
            a1 = ops.conv2d_bias_act_fwd(x, wf.data_ptr(), bf.data_ptr(), 64, 7, 7, 2, 3, relu=1)
        else:
            wpad = torch.empty((64, 7, 7, 4), dtype=F32, device=self.device)
            L.pad_channels(self._w(trunk.conv1.weight), wpad.data_ptr(), 64 * 49, cin, 4, 0, 0, st)
            c1 = ops.conv2d_fwd(x, wpad.data_ptr(), 64, 7, 7, 2, 3)
            # BN -> ReLU -> max-pool in one pass: the [N, 128, 128, 64] activation is never materialised
            bn = trunk.bn1
            stats = torch.empty(2, 64, dtype=F32, device=self.device)
            if train:
                ops.bn_stats(c1.numel() // 64, 64, c1, stats[0], stats[1], bn.running_mean.data_ptr(),
                             bn.running_var.data_ptr(), self._ws, bn.eps, bn.momentum)
            else:
                ops.bn_eval_prepare(bn.running_mean.data_ptr(), bn.running_var.data_ptr(), 64, stats[0], stats[1], bn.eps)
            st1 = (stats[0], stats[1])
            pool = ops.bn_relu_maxpool_bf16out if self._use16 else ops.bn_relu_maxpool
            p1, idx = pool(c1, st1[0], st1[1], self._w(bn.weight), self._w(bn.bias))
            return p1, (x, c1, None, st1, idx, cin)
        N, H1, W1, _ = a1.shape
        Ho, Wo = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        p1 = torch.empty((N, Ho, Wo, 64), dtype=F32, device=self.device)
        idx = torch.empty((N, Ho, Wo, 64), dtype=torch.uint8, device=self.device)
        L.maxpool3x3s2_fwd(a1.data_ptr(), p1.data_ptr(), idx.data_ptr(), N, H1, W1, 64, st)
        return p1, (x, c1, a1, st1, idx, cin)

    def _conv3x3(self, x, conv, K, stride):
        """3x3 conv of a BasicBlock: Winograd F(2x2, 3x3) where the shape allows it (stride 1, exact-fp32 mode: 2.25x
        fewer MFMA FLOPs - the direct kernel already runs at the chip's power-limited fp32 rate), else the implicit GEMM.
        -> (y, ud): ud = the transformed dgrad filter when the backward pass will want it (recording), else None."""
        if stride == 1 and self.use_winograd and ops.winograd_ok(x.shape, K):
            C = x.shape[-1]
            if self._recording and ops.winograd_ok((x.shape[0], x.shape[1], x.shape[2], K), C):
                u, ud = ops.winograd_weights(self._w(conv.weight), K, C, self.device, both=True)  # one launch for both
                return ops.conv3x3_winograd(x, u, K), ud
            u = ops.winograd_weights(self._w(conv.weight), K, C, self.device)
            return ops.conv3x3_winograd(x, u, K), None
        return ops.conv2d_fwd(x, self._w(conv.weight), K, 3, 3, stride, 1), None

    def _dgrad3x3(self, dy, conv, x_shape, stride, out=None, accumulate=False, ud=None):
        K = dy.shape[-1]
        if stride == 1 and self.use_winograd and ops.winograd_ok(dy.shape, x_shape[-1]):
            if ud is None:  # (weights are unchanged between forward and backward: normally handed over by the tape)
                ud = ops.winograd_weights(self._w(conv.weight), K, x_shape[-1], self.device, dgrad=True)
            return ops.conv3x3_winograd(dy, ud, x_shape[-1], out=out, accumulate=accumulate)
        return ops.conv2d_dgrad(dy, self._w(conv.weight), tuple(x_shape), 3, 3, stride, 1, out=out, accumulate=accumulate)

    def _bn_fwd16(self, bn, x, relu, residual, train):
        C = x.shape[-1]
        stats = torch.empty(2, C, dtype=F32, device=x.device)
        if train:
            ops.bf16_bn_stats(x.numel() // C, C, x, stats[0], stats[1], bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                              self._ws, bn.eps, bn.momentum)
        else:
            ops.bn_eval_prepare(bn.running_mean.data_ptr(), bn.running_var.data_ptr(), C, stats[0], stats[1], bn.eps)
        y = ops.bf16_bn_apply(x, stats[0], stats[1], self._w(bn.weight), self._w(bn.bias), relu, residual)
        return y, (stats[0], stats[1])

    def _block_fwd16(self, blk, x, train):
        """BasicBlock on bf16-stored feature maps: convs on csrc/bgemm.hip (direct implicit GEMM, bf16 tiles, bf16 weight
        shadow), BatchNorm reading / writing bf16 with fp32 statistics.  Same tape layout as _block_fwd."""
        K = blk.conv1.out_channels
        if train and self.fuse_bn_stats16:
            # train mode: the conv's epilogue emits the BatchNorm statistics of its (stored) output - no statistics pass
            def conv_bn(inp, conv, bn, R, stride, pad, relu, residual):
                stats = torch.empty(2, K, dtype=F32, device=inp.device)
                c = ops.bf16_conv2d_fwd_bnstats(inp, self._w16(conv.weight), K, R, R, stride, pad, stats[0], stats[1],
                                                bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self._ws, bn.eps,
                                                bn.momentum)
                y = ops.bf16_bn_apply(c, stats[0], stats[1], self._w(bn.weight), self._w(bn.bias), relu, residual)
                return c, y, (stats[0], stats[1])
            c1, a1, s1 = conv_bn(x, blk.conv1, blk.bn1, 3, blk.stride, 1, True, None)
            if blk.downsample is not None:
                cd, idn, sd = conv_bn(x, blk.downsample[0], blk.downsample[1], 1, blk.stride, 0, False, None)
            else:
                cd, sd, idn = None, None, x
            c2, out, s2 = conv_bn(a1, blk.conv2, blk.bn2, 3, 1, 1, True, idn)
            return out, (x, c1, a1, s1, c2, s2, cd, sd, out, None, None)
        c1 = ops.bf16_conv2d_fwd(x, self._w16(blk.conv1.weight), K, 3, 3, blk.stride, 1)
        a1, s1 = self._bn_fwd16(blk.bn1, c1, True, None, train)
        c2 = ops.bf16_conv2d_fwd(a1, self._w16(blk.conv2.weight), K, 3, 3, 1, 1)
        if blk.downsample is not None:
            cd = ops.bf16_conv2d_fwd(x, self._w16(blk.downsample[0].weight), K, 1, 1, blk.stride, 0)
            idn, sd = self._bn_fwd16(blk.downsample[1], cd, False, None, train)
        else:
            cd, sd, idn = None, None, x
        out, s2 = self._bn_fwd16(blk.bn2, c2, True, idn, train)
        return out, (x, c1, a1, s1, c2, s2, cd, sd, out, None, None)

    def _block_bwd16(self, blk, ctx, dout):
        x, c1, a1, s1, c2, s2, cd, sd, out, _, _ = ctx

        def bn_bwd(bn, dy, y_mask, xin, stats, want_dres=False, relu_no_residual=False):
            gw, aw = self._g(bn.weight)
            gb, _ = self._g(bn.bias)
            return ops.bf16_bn_bwd(dy, None if relu_no_residual else y_mask, xin, stats[0], stats[1], self._w(bn.weight), gw,
                                   gb, self._ws, want_dres=want_dres, accumulate=bool(aw),
                                   relu_beta_ptr=self._w(bn.bias) if relu_no_residual else 0)

        def wgrad(conv, xin, dy, R, stride, pad):
            gp, acc = self._g(conv.weight)
            self._wg_launch(lambda: ops.bf16_conv2d_wgrad(xin, dy, gp, R, R, stride, pad, self._ws, accumulate=bool(acc)),
                            (xin, dy))

        dc2, dres = bn_bwd(blk.bn2, dout, out, c2, s2, want_dres=True)
        wgrad(blk.conv2, a1, dc2, 3, 1, 1)
        da1 = ops.bf16_conv2d_dgrad(dc2, self._w16(blk.conv2.weight), tuple(a1.shape), 3, 3, 1, 1)
        dc1, _ = bn_bwd(blk.bn1, da1, a1, c1, s1, relu_no_residual=True)
        wgrad(blk.conv1, x, dc1, 3, blk.stride, 1)
        if blk.downsample is not None:
            dcd, _ = bn_bwd(blk.downsample[1], dres, None, cd, sd)
            wgrad(blk.downsample[0], x, dcd, 1, blk.stride, 0)
            dx = ops.bf16_conv2d_dgrad(dc1, self._w16(blk.conv1.weight), tuple(x.shape), 3, 3, blk.stride, 1)
            ops.bf16_conv2d_dgrad(dcd, self._w16(blk.downsample[0].weight), tuple(x.shape), 1, 1, blk.stride, 0, out=dx,
                                  accumulate=True)
        else:
            dx = dres
            ops.bf16_conv2d_dgrad(dc1, self._w16(blk.conv1.weight), tuple(x.shape), 3, 3, blk.stride, 1, out=dx,
                                  accumulate=True)
        return dx

    def _block_fwd(self, blk, x, train):
        if x.dtype == torch.bfloat16:
            return self._block_fwd16(blk, x, train)
        K = blk.conv1.out_channels
        if self._fold_now:
            # inference: eval-mode BN is an affine map per channel - folded into the conv weights, the block is three
            # (two) convolutions with bias / ReLU / identity epilogues and no BN pass at all
            Cin = x.shape[-1]

            def conv3(inp, conv, bn, stride, relu, residual=None):
                C = inp.shape[-1]
                wf, bf = ops.bn_fold(self._w(conv.weight), bn, K, 9, C)
                if stride == 1 and self.use_winograd and ops.winograd_ok(inp.shape, K):
                    u = ops.winograd_weights(wf.data_ptr(), K, C, self.device)
                    return ops.conv3x3_winograd_bias_act(inp, u, bf.data_ptr(), K, relu=relu, residual=residual)
                return ops.conv2d_bias_act_fwd(inp, wf.data_ptr(), bf.data_ptr(), K, 3, 3, stride, 1, relu=relu,
                                               residual=residual)

            a1 = conv3(x, blk.conv1, blk.bn1, blk.stride, 1)
            idn = x
            if blk.downsample is not None:
                wd, bd = ops.bn_fold(self._w(blk.downsample[0].weight), blk.downsample[1], K, 1, Cin)
                idn = ops.conv2d_bias_act_fwd(x, wd.data_ptr(), bd.data_ptr(), K, 1, 1, blk.stride, 0, relu=0)
            out = conv3(a1, blk.conv2, blk.bn2, 1, 2, residual=idn)
            return out, None
        c1, ud1 = self._conv3x3(x, blk.conv1, K, blk.stride)
        a1, s1 = self._bn_fwd(blk.bn1, c1, True, None, train)
        c2, ud2 = self._conv3x3(a1, blk.conv2, K, 1)
        if blk.downsample is not None:
            cd = ops.conv2d_fwd(x, self._w(blk.downsample[0].weight), K, 1, 1, blk.stride, 0)
            idn, sd = self._bn_fwd(blk.downsample[1], cd, False, None, train)
        else:
            cd, sd, idn = None, None, x
        C = c2.shape[-1]
        M = c2.numel() // C
        stats = torch.empty(2, C, dtype=F32, device=x.device)
        bn2 = blk.bn2
        if train:
            ops.bn_stats(M, C, c2, stats[0], stats[1], bn2.running_mean.data_ptr(), bn2.running_var.data_ptr(),
                         self._ws, bn2.eps, bn2.momentum)
        else:
            ops.bn_eval_prepare(bn2.running_mean.data_ptr(), bn2.running_var.data_ptr(), C, stats[0], stats[1], bn2.eps)
        out = ops.bn_apply(c2, stats[0], stats[1], self._w(bn2.weight), self._w(bn2.bias), True, idn)
        return out, (x, c1, a1, s1, c2, (stats[0], stats[1]), cd, sd, out, ud1, ud2)

    def _gpt_block_fwd16(self, blk, x, B, T, train):
        """_gpt_block_fwd on bf16-stored operands: LN -> h (bf16) -> fused k|q|v GEMM (bf16) -> attention on bf16 tiles (o
        bf16) -> proj GEMM + dropout + residual (fp32 stream) -> LN -> h2 (bf16) -> fc1 + ReLU (bf16) -> fc2 + dropout +
        residual (fp32).  Same dropout counters / masks as the fp32-storage path."""
        cfg = self.config
        C = x.shape[1]
        nh = cfg.n_head
        pa = cfg.attn_pdrop if train else 0.0
        pr = cfg.resid_pdrop if train else 0.0
        at = blk.attn
        h, m1, r1 = ops.layernorm_fwd_bf16(x, self._w(blk.ln1.weight), self._w(blk.ln1.bias), blk.ln1.eps)
        kqv = ops.bf16_linear_fwd(h, self._w16(at.key.weight), self._w(at.key.bias), 3 * C)   # bf16 [M, 3C]
        k, q, v = kqv[:, :C], kqv[:, C:2 * C], kqv[:, 2 * C:]
        off_a = self._next_drop(B * nh * T * T) if pa > 0 else 0
        y, lse = ops.attention_fwd_bf16(q, k, v, B, T, nh, self._ws, pa, self._seed, off_a)
        off_p = self._next_drop(x.numel()) if pr > 0 else 0
        x1 = ops.bf16_linear_fwd(y, self._w16(at.proj.weight), self._w(at.proj.bias), C, residual=x, drop_p=pr,
                                 seed=self._seed, seed_off=off_p)
        h2, m2, r2 = ops.layernorm_fwd_bf16(x1, self._w(blk.ln2.weight), self._w(blk.ln2.bias), blk.ln2.eps)
        fc1, fc2 = blk.mlp[0], blk.mlp[2]
        f1 = ops.bf16_linear_fwd(h2, self._w16(fc1.weight), self._w(fc1.bias), fc1.out_features, relu=True)
        off_m = self._next_drop(x.numel()) if pr > 0 else 0
        x2 = ops.bf16_linear_fwd(f1, self._w16(fc2.weight), self._w(fc2.bias), C, residual=x1, drop_p=pr, seed=self._seed,
                                 seed_off=off_m)
        return x2, (x, h, m1, r1, q, k, v, y, lse, off_a, pa, off_p, pr, x1, h2, m2, r2, f1, off_m)

    def _gpt_block_bwd16(self, blk, ctx, dx2, B, T, dz2, next_drop=None, want_dz=True):
        """backward of _gpt_block_fwd16.  dx2: fp32 gradient of the block output (residual stream); dz2: dropout(dx2) as
        bf16 (emitted by the LayerNorm backward above); returns (dx fp32, dropout(dx) as bf16 on the next block's mask)."""
        (x, h, m1, r1, q, k, v, y, lse, off_a, pa, off_p, pr, x1, h2, m2, r2, f1, off_m) = ctx
        C = x.shape[1]
        nh = self.config.n_head
        at = blk.attn
        fc1, fc2 = blk.mlp[0], blk.mlp[2]

        def wgrad(lin_w, lin_b, xin, dyin):
            gw, aw = self._g(lin_w)
            gb, _ = self._g(lin_b)
            self._wg_launch(lambda: ops.bf16_linear_wgrad(xin, dyin, gw, self._ws, accumulate=bool(aw), dbias_ptr=gb), (xin, dyin))

        wgrad(fc2.weight, fc2.bias, f1, dz2)
        df1 = ops.bf16_linear_dgrad(dz2, self._w16(fc2.weight), fc1.out_features, relu_mask_src=f1)
        wgrad(fc1.weight, fc1.bias, h2, df1)
        dh2 = ops.bf16_linear_dgrad(df1, self._w16(fc1.weight), C)
        g2w, a2 = self._g(blk.ln2.weight)
        g2b, _ = self._g(blk.ln2.bias)
        dx1, dz1 = ops.layernorm_bwd_bf16(dh2, x1, m2, r2, self._w(blk.ln2.weight), g2w, g2b, self._ws, add=dx2,
                                          accumulate=bool(a2), drop=(pr, self._seed, off_p))
        wgrad(at.proj.weight, at.proj.bias, y, dz1)
        dy = ops.bf16_linear_dgrad(dz1, self._w16(at.proj.weight), C)   # bf16: the attention backward's dO
        # the fused [3C, C] weight-gradient block and its [3C] bias block start at key.*: only valid while the three
        # projections' gradients are contiguous arena views with ONE accumulate flag (the fp32 path falls back to three
        # GEMMs otherwise; the bf16-storage path has no such fallback, so it refuses instead of writing wrong slices)
        if self._qkv_fused(at, grads=True) is None:
            raise RuntimeError("bf16-storage backward needs key / query / value gradients in the gradient arena with one "
                               "common state (all None or all arena views): call zero_grad(set_to_none=True) first")
        dkqv = torch.empty((dy.shape[0], 3 * C), dtype=torch.bfloat16, device=dy.device)
        ops.attention_bwd_bf16io(q, k, v, y, dy, lse, B, T, nh, self._attn_ws(B, T, nh, C), pa, self._seed, off_a,
                                 out=(dkqv[:, C:2 * C], dkqv[:, :C], dkqv[:, 2 * C:]))
        wgrad(at.key.weight, at.key.bias, h, dkqv)   # the fused [3C, C] block and its [3C] bias start at key.*
        dh = ops.bf16_linear_dgrad(dkqv, self._w16(at.key.weight), C)
        g1w, a1 = self._g(blk.ln1.weight)
        g1b, _ = self._g(blk.ln1.bias)
        return ops.layernorm_bwd_bf16(dh, x, m1, r1, self._w(blk.ln1.weight), g1w, g1b, self._ws, add=dx1,
                                      accumulate=bool(a1), drop=next_drop, want_drop=want_dz)

    def _gpt_block_fwd(self, blk, x, B, T, train):
        if self._use16:
            return self._gpt_block_fwd16(blk, x, B, T, train)
        cfg = self.config
        C = x.shape[1]
        nh = cfg.n_head
        pa = cfg.attn_pdrop if train else 0.0
        pr = cfg.resid_pdrop if train else 0.0
        at = blk.attn
        h, m1, r1 = ops.layernorm_fwd(x, self._w(blk.ln1.weight), self._w(blk.ln1.bias), blk.ln1.eps)
        fused = self._qkv_fused(at) if self.fuse_qkv else None
        if fused is not None:  # one GEMM, columns key | query | value
            kqv = ops.linear_fwd(h, fused[0], fused[1], 3 * C)
            k, q, v = kqv[:, :C], kqv[:, C:2 * C], kqv[:, 2 * C:]
        else:
            q = ops.linear_fwd(h, self._w(at.query.weight), self._w(at.query.bias), C)
            k = ops.linear_fwd(h, self._w(at.key.weight), self._w(at.key.bias), C)
            v = ops.linear_fwd(h, self._w(at.value.weight), self._w(at.value.bias), C)
        off_a = self._next_drop(B * nh * T * T) if pa > 0 else 0
        y, lse = ops.attention_fwd(q, k, v, B, T, nh, self._ws, pa, self._seed, off_a)
        off_p = self._next_drop(x.numel()) if pr > 0 else 0
        x1 = ops.linear_fwd(y, self._w(at.proj.weight), self._w(at.proj.bias), C, residual=x, drop_p=pr,
                            seed=self._seed, seed_off=off_p)
        h2, m2, r2 = ops.layernorm_fwd(x1, self._w(blk.ln2.weight), self._w(blk.ln2.bias), blk.ln2.eps)
        fc1, fc2 = blk.mlp[0], blk.mlp[2]
        f1 = ops.linear_fwd(h2, self._w(fc1.weight), self._w(fc1.bias), fc1.out_features, relu=True)
        off_m = self._next_drop(x.numel()) if pr > 0 else 0
        x2 = ops.linear_fwd(f1, self._w(fc2.weight), self._w(fc2.bias), C, residual=x1, drop_p=pr, seed=self._seed,
                            seed_off=off_m)
        return x2, (x, h, m1, r1, q, k, v, y, lse, off_a, pa, off_p, pr, x1, h2, m2, r2, f1, off_m)

    def _stage_fwd(self, s, feats, gps_src, B, train):
        """GPT fusion at scale s (1-based).  feats: 3 NHWC maps.  gps_src: (tensor, ptr, rows_per_group,
        group_stride, K) addressing the (B,2,K) input of vel_emb{s}."""
        L = lib()
        st = ops._stream()
        cfg = self.config
        S = cfg.seq_len
        C = STAGE_WIDTH[s - 1]
        gpt = getattr(self.encoder, f"transformer{s}")
        vel = getattr(self.encoder, f"vel_emb{s}")
        fps = (cfg.n_views * S, S, S)
        offs = (0, fps[0] * 64, (fps[0] + S) * 64)
        T = (cfg.n_views + 2) * S * 64 + 2
        assert gpt.pos_emb.shape == (1, T, C)
        pe = cfg.embd_pdrop if train else 0.0
        x0 = torch.empty((B, T, C), dtype=F32, device=self.device)
        off_e = self._next_drop(x0.numel()) if pe > 0 else 0
        pos = self._w(gpt.pos_emb)
        f16 = feats[0].dtype == torch.bfloat16   # bf16-storage path: feature maps bf16, tokens fp32
        for m in range(3):
            N, H = feats[m].shape[0], feats[m].shape[1]
            assert feats[m].shape == (B * fps[m], H, H, C) and (feats[m].dtype == torch.bfloat16) == f16
            (L.bf16_avgpool_tokens_fwd if f16 else L.avgpool_tokens_fwd)(
                feats[m].data_ptr(), pos, x0.data_ptr(), N, H, C, fps[m], offs[m], T, pe, self._seed, off_e, st)
        _, gptr, rpg, gstride, K = gps_src
        gemb = torch.empty((B, 2, C), dtype=F32, device=self.device)
        L.small_linear_fwd(gptr, self._w(vel.weight), self._w(vel.bias), gemb.data_ptr(), 2 * B, C, K, rpg, gstride,
                           0, st)
        L.gps_tokens_fwd(gemb.data_ptr(), pos, x0.data_ptr(), B, C, T, pe, self._seed, off_e, st)
        x = x0.view(B * T, C)
        blk_ctx = []
        for blk in gpt.blocks:
            x, c = self._gpt_block_fwd(blk, x, B, T, train)
            blk_ctx.append(c)
        xo, mf, rf = ops.layernorm_fwd(x, self._w(gpt.ln_f.weight), self._w(gpt.ln_f.bias), gpt.ln_f.eps)
        outs = []
        for m in range(3):
            N, H = feats[m].shape[0], feats[m].shape[1]
            o = torch.empty_like(feats[m])
            (L.bf16_upsample_add_fwd if f16 else L.upsample_add_fwd)(
                feats[m].data_ptr(), xo.data_ptr(), o.data_ptr(), N, H, C, fps[m], offs[m], T, st)
            outs.append(o)
        ctx = (s, C, T, fps, offs, pe, off_e, gps_src, blk_ctx, x, mf, rf, [f.shape for f in feats])
        return outs, xo, ctx

    def _dp_check_fresh_grads(self):
        # data parallel: a bucket is all-reduced in place the moment it is final, so every gradient must be written
        # fresh this step - accumulating into an already-reduced arena would sum the earlier steps world times over
        bad = [n for n, p in self._plist if p.grad is not None]
        if bad:
            raise RuntimeError(f"data-parallel backward needs zero_grad(set_to_none=True) first; {len(bad)} parameters "
                               f"still hold a gradient (e.g. {bad[0]}): gradient accumulation across steps is not "
                               "supported with the overlapped all-reduce")

    def _run_forward(self, images, lidars, radars, gps, record):
        """forward walk; the thread-local dropout-salt pointer of the library is scoped to the walk (set -> launches ->
        clear, also when a launch raises): no later launch of this thread can pick up a stale salt"""
        if record and self.grad_ready_hook is not None:
            self._dp_check_fresh_grads()   # before the forward advances the salt, the BN running stats and _nbt
        # parameter pointers of this walk: arena base + offset when every parameter still lives in the arena (one pass over
        # the list instead of ~1250 Tensor.data round trips per step); re-pointed parameters (EMA shadow applied) -> live reads
        self._wfast = self._wtable if (self._arena is not None and self.params_in_arena()) else None
        try:
            return self._run_forward_walk(images, lidars, radars, gps, record)
        finally:
            lib().set_dropout_salt(0)
            if not record:
                self._wfast = None   # a recorded forward keeps the table for its backward walk (cleared in _end_backward)

    def _run_forward_walk(self, images, lidars, radars, gps, record):
        L = lib()
        st = ops._stream()
        cfg = self.config
        train = self.training
        self._fold_now = self.fold_bn_eval and not train and not record  # inference only: backward needs the BN tape
        self._recording = bool(record)
        self._refresh_shadow16()
        if self._fold_now:
            self._use16 = False   # folded inference convs read the fp32 weights (BN folded per call)
        S = cfg.seq_len
        if torch.is_tensor(lidars):
            B = lidars.shape[0] // S
            assert images.shape[0] == B * S and radars.shape[0] == B * S
        else:
            B = lidars[0].shape[0]
            assert len(lidars) == S and len(radars) == S and len(images) == cfg.n_views * S
        assert gps.shape == (B, 2, 2), gps.shape
        if train:
            self._nbt.add_(1)
            self._advance_salt()
        L.set_dropout_salt(self._salt_cur.data_ptr() if train else 0)
        trunks = self._trunks()
        feats, stem_ctx = [], []
        streams = self._fork() if self.multi_stream else None
        for m, ((trunk, arch, cin, norm), frames) in enumerate(zip(trunks, (images, lidars, radars))):
            with self._trunk_ctx(streams, m):
                f, c = self._stem_fwd(trunk, cin, norm, frames, train)
            feats.append(f)
            stem_ctx.append(c)
        cap = getattr(self, "_capture", None)  # test hook: name -> list of NHWC / token tensors
        if cap is not None:
            if streams is not None:
                self._join()  # the stems ran on the trunk streams; the clone below is on the calling stream
            cap["stem"] = [f.clone() for f in feats]
        layer_ctx, stage_ctx = [], []
        gps_src = (gps, gps.data_ptr(), 2 * B, 0, 2)
        xo = None
        for s in range(1, 5):
            lc = []
            if streams is not None and s > 1:
                self._fork()
            for m, (trunk, arch, cin, norm) in enumerate(trunks):
                bc = []
                x = feats[m]
                with self._trunk_ctx(streams, m):
                    for blk in getattr(trunk, f"layer{s}"):
                        x, c = self._block_fwd(blk, x, train)
                        bc.append(c)
                feats[m] = x
                lc.append(bc)
            if streams is not None:
                self._join()
            layer_ctx.append(lc)
            if cap is not None:
                cap[f"layer{s}"] = [f.clone() for f in feats]
            feats, xo, sc = self._stage_fwd(s, feats, gps_src, B, train)
            if cap is not None:
                cap[f"gpt{s}"] = xo.clone()
                cap[f"fused{s}"] = [f.clone() for f in feats]
            stage_ctx.append(sc)
            C, T = sc[1], sc[2]
            gps_src = (xo, xo.data_ptr() + (T - 2) * C * 4, 2, T * C, C)
        # head: global pool, 17-token sum, join MLP
        C, T = 512, stage_ctx[-1][2]
        pooled = []
        for m in range(3):
            N = feats[m].shape[0]
            assert feats[m].shape[1:] == (8, 8, 512)
            pl = torch.empty((N, 512), dtype=F32, device=self.device)
            (L.bf16_global_pool if feats[m].dtype == torch.bfloat16 else L.global_pool)(feats[m].data_ptr(), pl.data_ptr(), N, 512, st)
            pooled.append(pl)
        fused = torch.empty((B, 512), dtype=F32, device=self.device)
        L.head_sum(pooled[0].data_ptr(), pooled[1].data_ptr(), pooled[2].data_ptr(), xo.data_ptr(), fused.data_ptr(), B,
                   512, cfg.n_views * S, S, T, st)
        if cap is not None:
            cap["fused"] = fused.clone()
        j0, j2, j4 = self.join[0], self.join[2], self.join[4]
        h1 = torch.empty((B, 256), dtype=F32, device=self.device)
        h2 = torch.empty((B, 128), dtype=F32, device=self.device)
        logits = torch.empty((B, 64), dtype=F32, device=self.device)
        L.small_linear_fwd(fused.data_ptr(), self._w(j0.weight), self._w(j0.bias), h1.data_ptr(), B, 256, 512, B, 0, 1, st)
        L.small_linear_fwd(h1.data_ptr(), self._w(j2.weight), self._w(j2.bias), h2.data_ptr(), B, 128, 256, B, 0, 1, st)
        L.small_linear_fwd(h2.data_ptr(), self._w(j4.weight), self._w(j4.bias), logits.data_ptr(), B, 64, 128, B, 0, 0, st)
        gru = None
        if self.gru_head:  # model2_seq_30to5.py:846-862: logits is the GRU's initial hidden state
            T = self.pred_len
            pred = torch.empty((B, T, 64), dtype=F32, device=self.device)
            saved = torch.empty(L.gru_head_saved_floats(B, T), dtype=F32, device=self.device) if record else None
            d = self.decoder
            L.gru_head_fwd(logits.data_ptr(), self._w(d.weight_ih), self._w(d.weight_hh), self._w(d.bias_ih),
                           self._w(d.bias_hh), self._w(self.output.weight), self._w(self.output.bias), pred.data_ptr(),
                           0 if saved is None else saved.data_ptr(), B, T, 64, st)
            gru = (logits, saved)
            logits = pred
        tape = None
        if record:
            tape = (B, stem_ctx, layer_ctx, stage_ctx, (fused, h1, h2, [f.shape for f in feats], gru, feats[0].dtype), gps,
                    self._salt_cur if train else None)
        return logits, tape

    # ================================================================ backward walk =============
    def _wgrad_conv(self, conv, x, dy, R, stride, pad):
        gp, acc = self._g(conv.weight)
        K = dy.shape[-1]
        if (R == 3 and stride == 1 and self.use_winograd and x.shape[-1] * K >= 128 * 128
                and ops.winograd_wgrad_ok(x.shape, K)):
            # Winograd-domain weight gradient (measured faster from 128 x 128 channels up; 64 x 64 stays direct)
            self._wg_launch(lambda: ops.conv3x3_winograd_wgrad(x, dy, gp, self._ws, accumulate=bool(acc)), (x, dy))
            return
        self._wg_launch(lambda: ops.conv2d_wgrad(x, dy, gp, R, R, stride, pad, self._ws, accumulate=bool(acc)), (x, dy))

    # Weight gradients feed nothing but the optimizer, while the dgrad / attention / LayerNorm kernels around them form
    # the serial chain of the backward walk.  In the GPT stages (one stream) the weight-gradient launches therefore go
    # to a companion stream, ordered after the producer of dy and joined at the end of the stage: they overlap the
    # chain and fill its launch tails (+3.8 % step throughput).  The same for the trunk streams (overlap_wgrad_trunks)
    # was measured and gains nothing on top of the three concurrent trunks.
    def _wg_launch(self, fn, keep):
        if not (self.multi_stream and self.overlap_wgrad):
            fn()
            return
        cur = ops.current_stream_obj()
        if not self.overlap_wgrad_trunks and cur.cuda_stream in self._ws_side:  # a trunk stream
            fn()
            return
        side = self._wg_map.get(cur.cuda_stream)
        if side is None:
            side = torch.cuda.Stream(self.device)
            self._wg_map[cur.cuda_stream] = side
            self._ws_side[side.cuda_stream] = ops.Workspace(self.device, 256 << 20)
        side.wait_stream(cur)
        with ops.on_stream(side):
            fn()  # self._ws resolves to the companion stream's own scratch
        self._wg_used[side.cuda_stream] = side
        self._wg_keep.append(keep)  # dy / x must outlive the launch on the other stream

    def _wg_join(self):
        """the calling stream waits for every outstanding weight-gradient launch (gradients final after this)"""
        if self._wg_used:
            cur = ops.current_stream_obj()
            for side in self._wg_used.values():
                cur.wait_stream(side)
            self._wg_used = {}
        self._wg_keep = []

    def _bn_bwd(self, bn, dy, y_mask, x, stats, want_dres=False, relu_no_residual=False):
        """relu_no_residual: y_mask is relu(bn(x)) itself (bn1 of a block / of the stem) - its sign is recomputed from
        x inside the kernels instead of reading the activation tensor twice"""
        gw, aw = self._g(bn.weight)
        gb, ab = self._g(bn.bias)
        if relu_no_residual:
            dx, dres = ops.bn_bwd(dy, None, x, stats[0], stats[1], self._w(bn.weight), gw, gb, self._ws,
                                  want_dres=want_dres, accumulate=bool(aw), relu_beta_ptr=self._w(bn.bias))
            return dx, dres
        dx, dres = ops.bn_bwd(dy, y_mask, x, stats[0], stats[1], self._w(bn.weight), gw, gb, self._ws,
                              want_dres=want_dres, accumulate=bool(aw))
        return dx, dres

    def _lin_param_grads(self, lin, x, dy):
        gw, aw = self._g(lin.weight)
        gb, ab = self._g(lin.bias)
        self._linear_wgrad(x, dy, gw, bool(aw), gb)

    def _linear_wgrad(self, x, dy, gw, accumulate, gb):
        self._wg_launch(lambda: ops.linear_wgrad(x, dy, gw, self._ws, accumulate=accumulate, dbias_ptr=gb), (x, dy))

    def _block_bwd(self, blk, ctx, dout, need_dx=True):
        if dout.dtype == torch.bfloat16:
            return self._block_bwd16(blk, ctx, dout)
        x, c1, a1, s1, c2, s2, cd, sd, out, ud1, ud2 = ctx
        dc2, dres = self._bn_bwd(blk.bn2, dout, out, c2, s2, want_dres=True)
        self._wgrad_conv(blk.conv2, a1, dc2, 3, 1, 1)
        da1 = self._dgrad3x3(dc2, blk.conv2, a1.shape, 1, ud=ud2)
        dc1, _ = self._bn_bwd(blk.bn1, da1, a1, c1, s1, relu_no_residual=True)
        self._wgrad_conv(blk.conv1, x, dc1, 3, blk.stride, 1)
        if blk.downsample is not None:
            dcd, _ = self._bn_bwd(blk.downsample[1], dres, None, cd, sd)
            self._wgrad_conv(blk.downsample[0], x, dcd, 1, blk.stride, 0)
            # the 3x3 dgrad writes every input pixel; the strided 1x1 only touches the even/even parity class
            dx = self._dgrad3x3(dc1, blk.conv1, x.shape, blk.stride, ud=ud1)
            ops.conv2d_dgrad(dcd, self._w(blk.downsample[0].weight), tuple(x.shape), 1, 1, blk.stride, 0, out=dx,
                             accumulate=True)
        else:
            dx = dres
            self._dgrad3x3(dc1, blk.conv1, x.shape, blk.stride, out=dx, accumulate=True, ud=ud1)
        return dx

    def _gpt_block_bwd(self, blk, ctx, dx2, B, T, dz2=None, next_drop=None):
        """dx2: gradient of the block output; dz2: dropout(dx2) on this block's resid_drop mask if the producer of
        dx2 already emitted it (fused into its LayerNorm backward); next_drop = (p, seed, off) of the block below:
        the final LayerNorm backward then also emits dropout(dx).  Returns (dx, dropout(dx) or None)."""
        (x, h, m1, r1, q, k, v, y, lse, off_a, pa, off_p, pr, x1, h2, m2, r2, f1, off_m) = ctx
        C = x.shape[1]
        nh = self.config.n_head
        at = blk.attn
        fc1, fc2 = blk.mlp[0], blk.mlp[2]
        # x2 = x1 + drop(fc2(f1))
        if dz2 is None:
            dz2 = ops.dropout(dx2, pr, self._seed, off_m) if pr > 0 else dx2
        self._lin_param_grads(fc2, f1, dz2)
        df1 = ops.linear_dgrad(dz2, self._w(fc2.weight), fc1.out_features, relu_mask_src=f1)
        self._lin_param_grads(fc1, h2, df1)
        dh2 = ops.linear_dgrad(df1, self._w(fc1.weight), C)
        g2w, a2 = self._g(blk.ln2.weight)
        g2b, _ = self._g(blk.ln2.bias)
        # x1 = x + drop(proj(y)): the LayerNorm backward emits dx1 and dropout(dx1) together
        dx1, dz1 = ops.layernorm_bwd(dh2, x1, m2, r2, self._w(blk.ln2.weight), g2w, g2b, self._ws, add=dx2,
                                     accumulate=bool(a2), drop=(pr, self._seed, off_p))
        self._lin_param_grads(at.proj, y, dz1)
        dy = ops.linear_dgrad(dz1, self._w(at.proj.weight), C)
        fw = self._qkv_fused(at) if self.fuse_qkv else None
        fg = self._qkv_fused(at, grads=True) if fw is not None else None
        if fg is not None:  # gradients of the fused projection: one [M, 3C] matrix, one wgrad, one dgrad
            dkqv = torch.empty((dy.shape[0], 3 * C), dtype=F32, device=dy.device)
            ops.attention_bwd(q, k, v, y, dy, lse, B, T, nh, self._attn_ws(B, T, nh, C), pa, self._seed, off_a,
                              out=(dkqv[:, C:2 * C], dkqv[:, :C], dkqv[:, 2 * C:]))
            self._linear_wgrad(h, dkqv, fg[0], bool(self._g(at.key.weight)[1]), fg[1])
            dh = ops.linear_dgrad(dkqv, fw[0], C)
        else:
            dq, dk, dv = ops.attention_bwd(q, k, v, y, dy, lse, B, T, nh, self._attn_ws(B, T, nh, C), pa, self._seed,
                                           off_a)
            self._lin_param_grads(at.query, h, dq)
            self._lin_param_grads(at.key, h, dk)
            self._lin_param_grads(at.value, h, dv)
            dh = ops.linear_dgrad(dq, self._w(at.query.weight), C)
            ops.linear_dgrad(dk, self._w(at.key.weight), C, out=dh, accumulate=True)
            ops.linear_dgrad(dv, self._w(at.value.weight), C, out=dh, accumulate=True)
        g1w, a1 = self._g(blk.ln1.weight)
        g1b, _ = self._g(blk.ln1.bias)
        if next_drop is not None:
            return ops.layernorm_bwd(dh, x, m1, r1, self._w(blk.ln1.weight), g1w, g1b, self._ws, add=dx1,
                                     accumulate=bool(a1), drop=next_drop)
        dx = ops.layernorm_bwd(dh, x, m1, r1, self._w(blk.ln1.weight), g1w, g1b, self._ws, add=dx1,
                               accumulate=bool(a1))
        return dx, None

    def _stage_bwd(self, ctx, dfeats_out, dgps_tok, B):
        """dfeats_out: grads of the 3 post-fusion maps; dgps_tok: (tensor, bcast) grad of the GPS rows of
        this stage's output.  Returns grads of the 3 pre-fusion maps and leaves the GPS-input gradient
        in self._dgps_prev."""
        L = lib()
        st = ops._stream()
        (s, C, T, fps, offs, pe, off_e, gps_src, blk_ctx, x_last, mf, rf, fshapes) = ctx
        gpt = getattr(self.encoder, f"transformer{s}")
        vel = getattr(self.encoder, f"vel_emb{s}")
        dxo = torch.empty((B * T, C), dtype=F32, device=self.device)
        f16 = dfeats_out[0].dtype == torch.bfloat16
        for m in range(3):
            N, H = fshapes[m][0], fshapes[m][1]
            (L.bf16_upsample_add_bwd if f16 else L.upsample_add_bwd)(dfeats_out[m].data_ptr(), dxo.data_ptr(), N, H, C, fps[m],
                                                                     offs[m], T, st)
        gsrc, bcast = dgps_tok
        L.gps_rows(gsrc.data_ptr(), dxo.data_ptr(), B, C, T, 1, 0, int(bcast), st)
        gfw, af = self._g(gpt.ln_f.weight)
        gfb, _ = self._g(gpt.ln_f.bias)
        rev = list(zip(reversed(list(gpt.blocks)), reversed(blk_ctx)))
        drops = [(bc[12], self._seed, bc[18]) for _, bc in rev]  # (resid_pdrop, seed, fc2-branch mask offset) per block
        if blk_ctx and blk_ctx[0][1].dtype == torch.bfloat16:   # recorded on the bf16-storage path
            dx, dz = ops.layernorm_bwd_bf16(dxo, x_last, mf, rf, self._w(gpt.ln_f.weight), gfw, gfb, self._ws,
                                            accumulate=bool(af), drop=drops[0])
            for i, (blk, bc) in enumerate(rev):
                last = i + 1 == len(rev)
                dx, dz = self._gpt_block_bwd16(blk, bc, dx, B, T, dz, next_drop=None if last else drops[i + 1],
                                               want_dz=not last)
        else:
            dx, dz = ops.layernorm_bwd(dxo, x_last, mf, rf, self._w(gpt.ln_f.weight), gfw, gfb, self._ws,
                                       accumulate=bool(af), drop=drops[0])
            for i, (blk, bc) in enumerate(rev):
                dx, dz = self._gpt_block_bwd(blk, bc, dx, B, T, dz2=dz, next_drop=drops[i + 1] if i + 1 < len(rev) else None)
        self._wg_join()
        dpre = ops.dropout(dx, pe, self._seed, off_e) if pe > 0 else dx
        gpos, apos = self._g(gpt.pos_emb)
        L.batch_sum(dpre.data_ptr(), gpos, T * C, B, T * C, apos, st)
        dfeats = []
        for m in range(3):
            N, H = fshapes[m][0], fshapes[m][1]
            d = torch.empty(tuple(fshapes[m]), dtype=dfeats_out[m].dtype, device=self.device)
            (L.bf16_avgpool_tokens_bwd if f16 else L.avgpool_tokens_bwd)(dpre.data_ptr(), dfeats_out[m].data_ptr(), d.data_ptr(),
                                                                       N, H, C, fps[m], offs[m], T, st)
            dfeats.append(d)
        dgemb = torch.empty((B, 2, C), dtype=F32, device=self.device)
        L.gps_rows(dpre.data_ptr(), dgemb.data_ptr(), B, C, T, 0, 0, 0, st)
        src_t, gptr, rpg, gstride, K = gps_src
        gw, aw = self._g(vel.weight)
        gb, _ = self._g(vel.bias)
        if s > 1:
            dprev = torch.empty((B, 2, K), dtype=F32, device=self.device)
            dptr = dprev.data_ptr()
        else:
            dprev, dptr = None, 0
        L.small_linear_bwd(dgemb.data_ptr(), 0, gptr, self._w(vel.weight), dptr, gw, gb, 2 * B, C, K, rpg, gstride,
                           2 * B, 0, 0, aw, st)
        return dfeats, dprev

    def _stem_bwd(self, trunk, ctx, dpool, cin):
        L = lib()
        st = ops._stream()
        x, c1, _, st1, idx, _ = ctx
        bn = trunk.bn1
        gw_bn, a_bn = self._g(bn.weight)
        gb_bn, _ = self._g(bn.bias)
        if c1.dtype == torch.bfloat16:   # the bf16 stem (csrc/stem.hip)
            dc1 = ops.bf16_stem_bn_bwd_maxpool(dpool, idx, c1, st1[0], st1[1], self._w(bn.weight), self._w(bn.bias), gw_bn,
                                               gb_bn, self._ws, accumulate=bool(a_bn))
            gw, aw = self._g(trunk.conv1.weight)
            self._wg_launch(lambda: ops.bf16_stem_wgrad(x, dc1, gw, cin, self._ws, accumulate=bool(aw)), (x, dc1))
            return
        bwd_pool = ops.bn_bwd_maxpool_bf16in if dpool.dtype == torch.bfloat16 else ops.bn_bwd_maxpool
        dc1 = bwd_pool(dpool, idx, c1, st1[0], st1[1], self._w(bn.weight), self._w(bn.bias), gw_bn, gb_bn,
                       self._ws, accumulate=bool(a_bn))
        dwpad = torch.empty((64, 7, 7, 4), dtype=F32, device=self.device)
        gw, aw = self._g(trunk.conv1.weight)

        def stem_wgrad():
            ops.conv2d_wgrad(x, dc1, dwpad.data_ptr(), 7, 7, 2, 3, self._ws)
            L.pad_channels(dwpad.data_ptr(), gw, 64 * 49, cin, 4, 1, aw, ops._stream())
        self._wg_launch(stem_wgrad, (x, dc1, dwpad))

    def _run_backward(self, tape, dlogits):
        try:
            self._run_backward_walk(tape, dlogits)
        finally:
            lib().set_dropout_salt(0)
            self._wfast = None

    def _run_backward_walk(self, tape, dlogits):
        L = lib()
        st = ops._stream()
        cfg = self.config
        B, stem_ctx, layer_ctx, stage_ctx, head, gps, salt = tape
        fused, h1, h2, fshapes, gru, fdtype = head
        L.set_dropout_salt(salt.data_ptr() if salt is not None else 0)
        self._begin_backward()
        if gru is not None:  # back through the GRU head: dpred (B, pred_len, 64) -> gradient of the join output
            z0, saved = gru
            T = self.pred_len
            assert dlogits.shape == (B, T, 64) and dlogits.dtype == F32
            npar = L.gru_head_slab_floats()
            slabs = torch.empty((B, npar), dtype=F32, device=self.device)
            dz = torch.empty((B, 64), dtype=F32, device=self.device)
            d = self.decoder
            L.gru_head_bwd(dlogits.contiguous().data_ptr(), z0.data_ptr(), saved.data_ptr(), self._w(d.weight_ih),
                           self._w(d.weight_hh), self._w(self.output.weight), dz.data_ptr(), slabs.data_ptr(), B, T, 64, st)
            off = 0
            for prm in (d.weight_ih, d.weight_hh, d.bias_ih, d.bias_hh, self.output.weight, self.output.bias):
                gp, acc = self._g(prm)
                L.batch_sum(slabs.data_ptr() + 4 * off, gp, prm.numel(), B, npar, acc, st)
                off += prm.numel()
            dlogits = dz
        assert dlogits.shape == (B, 64) and dlogits.dtype == F32
        S = cfg.seq_len
        j0, j2, j4 = self.join[0], self.join[2], self.join[4]
        dh2 = torch.empty_like(h2)
        dh1 = torch.empty_like(h1)
        dfused = torch.empty_like(fused)

        def small_bwd(lin, dy, ymask, x, dx, M, N, K):
            gw, aw = self._g(lin.weight)
            gb, _ = self._g(lin.bias)
            L.small_linear_bwd(dy.data_ptr(), 0 if ymask is None else ymask.data_ptr(), x.data_ptr(),
                               self._w(lin.weight), dx.data_ptr(), gw, gb, M, N, K, M, 0, M, 0, 0, aw, st)

        small_bwd(j4, dlogits, None, h2, dh2, B, 64, 128)
        small_bwd(j2, dh2, h2, h1, dh1, B, 128, 256)
        small_bwd(j0, dh1, h1, fused, dfused, B, 256, 512)
        self._milestone_done(0)
        trunks = self._trunks()
        dfeats = []
        for m in range(3):
            d = torch.empty(tuple(fshapes[m]), dtype=fdtype, device=self.device)
            fps = cfg.n_views * S if m == 0 else S
            (L.bf16_head_bwd if fdtype == torch.bfloat16 else L.head_bwd)(dfused.data_ptr(), d.data_ptr(), fshapes[m][0], 512,
                                                                        fps, st)
            dfeats.append(d)
        dgps = (dfused, True)
        for s in range(4, 0, -1):
            dfeats, dprev = self._stage_bwd(stage_ctx[s - 1], dfeats, dgps, B)
            self._milestone_done(1 + 2 * (4 - s))
            dgps = (dprev, False)
            streams = self._fork() if self.multi_stream else None
            for m, (trunk, arch, cin, norm) in enumerate(trunks):
                blocks = list(getattr(trunk, f"layer{s}"))
                d = dfeats[m]
                with self._trunk_ctx(streams, m):
                    for blk, bc in zip(reversed(blocks), reversed(layer_ctx[s - 1][m])):
                        d = self._block_bwd(blk, bc, d)
                    if s == 1:  # the stem backward continues on the same trunk stream
                        self._stem_bwd(trunk, stem_ctx[m], d, cin)
                dfeats[m] = d
            if streams is not None:
                self._join()
            self._wg_join()
            self._milestone_done(2 + 2 * (4 - s))
        self._milestone_done(9)
        self._end_backward()


class TransFuser30to5(TransFuser):
    """Drop-in for /root/reference/model2_seq_30to5.py::TransFuser (:831-862) with the GPT encoder: the same fusion
    path at ``config.seq_len`` = 10 (1922 tokens) followed by ``decoder = nn.GRUCell(64, 64)`` / ``output =
    nn.Linear(64, 64)`` unrolled ``config.pred_len`` (5) times; ``forward`` returns (B, pred_len, 64)."""
    _GRU_HEAD = True

