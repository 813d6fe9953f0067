"""Training-step pieces around the fusion model, mirroring /root/reference/train2_seq.py:
FocalLoss (:291-301), AdamW (:539) and EMA (:303-334) as fused HIP kernels over the flat
parameter arena, the per-epoch cyclic-cosine LR schedule (scheduler.py:82-119 with the arguments of
train2_seq.py:541-547), the top-k / DBA metrics (:347-383) and the per-iteration order of
Engine.train (:94-157): zero_grad -> forward -> focal(soft target) -> backward -> step -> ema.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from . import ops
from ._lib import lib

F32 = torch.float32


# ------------------------------------------------------------------------------------------------
class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, alpha, gamma):
        logits = logits.contiguous()
        target = target.contiguous()
        if tuple(target.shape) != tuple(logits.shape) or target.dtype != F32 or target.device != logits.device:
            raise ValueError(f"focal-loss target {tuple(target.shape)}/{target.dtype} does not match logits "
                             f"{tuple(logits.shape)}/fp32 (the kernel reads logits.numel() floats from both)")
        loss = torch.empty(1, dtype=F32, device=logits.device)
        dlogits = torch.empty_like(logits)
        lib().focal_loss(logits.data_ptr(), target.data_ptr(), loss.data_ptr(), dlogits.data_ptr(), logits.numel(),
                         alpha, gamma, 1.0, ops._stream())
        ctx.save_for_backward(dlogits)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        out = torch.empty_like(dlogits)
        # dlogits * upstream through the library's axpby kernel (beta term unused)
        if g.numel() == 1 and dlogits.numel() % 4 == 0:
            lib().axpby(dlogits.data_ptr(), 0, out.data_ptr(), dlogits.numel(), float(g), 0.0, ops._stream())
        else:  # pragma: no cover
            raise RuntimeError("unexpected focal-loss upstream gradient")
        return out, None, None, None


class FocalLoss(nn.Module):
    """train2_seq.py:291-301 (sigmoid focal loss, alpha .25, gamma 2, mean; 1-D integer targets are
    one-hot encoded to 64 classes)."""

    def __init__(self, gamma=2, alpha=0.25):
        super().__init__()
        self.gamma = gamma
        self.alpha = alpha

    def forward(self, input, target):
        if target.dim() == 1:
            target = torch.nn.functional.one_hot(target.long(), num_classes=64)
        return _FocalFn.apply(input, target.to(input.device, F32), float(self.alpha), float(self.gamma))


# ------------------------------------------------------------------------------------------------
class FusedAdamW:
    """torch.optim.AdamW(model.parameters(), lr) semantics (train2_seq.py:539: betas (.9,.999), eps 1e-8,
    weight_decay .01 on every parameter) as ONE streaming kernel over the parameter arena, optionally
    fused with the EMA shadow update (train2_seq.py:315-320)."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ema_decay=None,
                 max_grad_norm=None):
        """max_grad_norm: torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) of the 30->5 training step
        (train2_seq_30to5.py:120, max_norm 3.0) folded into the step: one reduction over the gradient arena, the clip
        coefficient stays on the device and is multiplied into the AdamW kernel's gradient scale (no host sync, the
        arena is not rewritten)."""
        self.model = model
        self.max_grad_norm = max_grad_norm
        self._clip_out = torch.zeros(2, dtype=F32, device=model.device)           # [total norm, clip coefficient]
        self._clip_ws = torch.zeros(8 + 8 * 1024, dtype=torch.uint8, device=model.device)
        # per-step scalars in device memory {lr, bc1, bc2_sqrt, step}: the step is hipGraph-replayable (CapturedTrainStep)
        self._dev = torch.zeros(4, dtype=F32, device=model.device)
        self._dev_lr = None
        p, g = model.flat_parameters()
        self.m = torch.zeros_like(p)
        self.v = torch.zeros_like(p)
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.step_count = 0
        self.ema_decay = ema_decay
        self.shadow = p.clone() if ema_decay is not None else None
        self.grad_scale = 1.0  # 1/world_size under data parallelism (sum all-reduce, scale here)
        self._overlap = None   # (reducer, side stream) when enable_overlap() is active
        self._began = False    # this step's optimizer scalars were already advanced by the overlapped path
        self._applied = 0      # arena prefix already updated by the overlapped buckets of the backward walk in flight

    # ---- the update overlapped with the backward walk -------------------------------------------------------------------
    def enable_overlap(self, model=None, min_bucket_elems=8 << 20):
        """AdamW (+EMA) is one HBM-bound pass over 2.2 GB (0.4 ms at bs 12) that the reference runs after backward
        (train2_seq.py:128-134).  The gradient arena is laid out in backward-completion order, so the parameters of a bucket
        can be updated the moment the bucket's gradients are final - on a side stream, under the matrix-core-bound rest of
        the backward walk (under data parallelism: the moment the bucket's all-reduce has completed).  step() then only joins
        that stream and updates whatever was not covered.  Same arithmetic, same result bit for bit
        (tests/test_train_gpu.py); a parameter is never updated while a backward kernel may still read it: a bucket is
        final only after every kernel of its layers was enqueued and the producing streams were joined.  Not available with
        max_grad_norm (the clip coefficient needs the norm of the whole gradient first).  Call after dist.attach() when
        both are used."""
        from . import dist as ddist
        model = model or self.model
        if self.max_grad_norm is not None:
            raise RuntimeError("enable_overlap(): a global-norm clip needs the whole gradient before the first update")
        red = getattr(model.grad_ready_hook, "__self__", None) if model.grad_ready_hook is not None else None
        if red is None:   # single process: the reducer is only the bucketing engine (no collective at world size 1)
            red = ddist.GradReducer(model.flat_parameters()[1], None, min_bucket_elems, flush_at=7)
            model.grad_ready_hook = red.ready
        red.on_begin, red.on_bucket = self._overlap_begin, self._overlap_bucket
        self._overlap = (red, torch.cuda.Stream(model.device))
        return red

    def _overlap_begin(self):
        # start of a backward walk: this step's scalars (step count, bias corrections, lr) before its first bucket.  The
        # reducer's begin() may run more than once per iteration (train_iteration and the model's backward both call it):
        # the state advances once per step()
        self._applied = 0
        if not self._began:
            self._began = True
            self._advance_state()

    def _overlap_bucket(self, lo, hi, work):
        _, side = self._overlap
        side.wait_stream(torch.cuda.current_stream())   # the bucket is final on the calling stream (producers joined)
        with torch.cuda.stream(side):
            if work is not None:
                work.wait()                             # ... and summed over the ranks
            self._update_range(lo, hi)
        assert lo == self._applied
        self._applied = hi

    def _advance_state(self):
        self.step_count += 1
        grp = self.param_groups[0]
        self.sync_lr()
        lib().adamw_state_advance(self._dev.data_ptr(), grp["betas"][0], grp["betas"][1], ops._stream())

    def _update_range(self, lo, hi, coef=0):
        p, g = self.model.flat_parameters()
        grp = self.param_groups[0]
        n = hi - lo
        lib().adamw_step_dev(p.data_ptr() + 4 * lo, g.data_ptr() + 4 * lo, self.m.data_ptr() + 4 * lo, self.v.data_ptr() + 4 * lo,
                             0 if self.shadow is None else self.shadow.data_ptr() + 4 * lo, n, self._dev.data_ptr(),
                             grp["betas"][0], grp["betas"][1], grp["eps"], grp["weight_decay"],
                             0.0 if self.ema_decay is None else float(self.ema_decay), float(self.grad_scale), coef,
                             ops._stream())

    def zero_grad(self, set_to_none=True):
        if self._began:
            # enable_overlap(): a backward walk already advanced this step's scalars and updated its buckets, and no step()
            # followed (backward == a partial step while overlap is on).  Starting the next iteration on stale scalars would
            # silently skip an _advance_state(): refuse instead.
            raise RuntimeError("FusedAdamW.zero_grad(): a backward ran under enable_overlap() without a following step(); "
                               "with overlap on, backward already applies the update bucket by bucket - call step() after it")
        if set_to_none:
            for _, p in getattr(self.model, "_plist", None) or self.model.named_parameters():
                p.grad = None
        else:
            self.model.flat_parameters()[1].zero_()

    def step(self):
        if not self.model.params_in_arena():
            raise RuntimeError("parameters were re-pointed away from the arena (EMA shadow applied?): restore first")
        p, g = self.model.flat_parameters()
        if self._overlap is not None and self._began:
            # the backward walk already advanced the step's scalars and updated the prefix [0, _applied) bucket by bucket on
            # the side stream: join it and finish what is left (nothing when every bucket went through the hook)
            torch.cuda.current_stream().wait_stream(self._overlap[1])
            if self._applied < p.numel():
                self._update_range(self._applied, p.numel())
            self._applied, self._began = 0, False
            return
        self._advance_state()
        coef = 0
        if self.max_grad_norm is not None:
            lib().grad_norm_clip(g.data_ptr(), g.numel(), float(self.max_grad_norm), float(self.grad_scale),
                                 self._clip_out.data_ptr(), self._clip_ws.data_ptr(), self._clip_ws.numel(), ops._stream())
            coef = self._clip_out.data_ptr() + 4
        self._update_range(0, p.numel(), coef)

    def sync_lr(self):
        """write the scheduled lr to device memory when it changed (a tiny copy, outside any captured graph)"""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._dev_lr:
            self._dev[0:1].fill_(lr)
            self._dev_lr = lr

    def last_grad_norm(self):
        """total gradient norm of the last clipped step (host read: synchronises)"""
        return float(self._clip_out[0].item())

    def state_dict(self):
        return dict(m=self.m, v=self.v, step=self.step_count, shadow=self.shadow, param_groups=self.param_groups,
                    dropout_rng=self.model.rng_state())

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
        self._dev.view(torch.int32)[3:4].fill_(self.step_count)
        if self.shadow is not None and sd.get("shadow") is not None:
            self.shadow.copy_(sd["shadow"])
        if sd.get("dropout_rng") is not None:  # resume the mask sequence, do not replay it from step 0
            self.model.set_rng_state(sd["dropout_rng"])


class EMA:
    """train2_seq.py:303-334 API (register / update / apply_shadow / restore) over a shadow arena.
    apply_shadow re-points every ``param.data`` at its shadow view exactly as the reference does
    (:326-327); the kernels read parameter pointers at call time, so evaluation then runs on the
    averaged weights with no copy."""

    def __init__(self, model, decay, optimizer: FusedAdamW | None = None):
        self.model = model
        self.decay = decay
        self.optimizer = optimizer
        self.shadow = None
        self.backup = {}
        self._views = None

    def register(self):
        if self.optimizer is not None and self.optimizer.shadow is not None:
            self.shadow = self.optimizer.shadow  # updated inside the fused optimizer kernel
        else:
            self.shadow = self.model.flat_parameters()[0].clone()
        self._views = {}
        for name, p in self.model.named_parameters():
            off, n = self.model._pslice[name]
            seg = self.shadow[off:off + n]
            if p.dim() == 4:
                O, I, R, S = p.shape
                self._views[name] = seg.view(O, R, S, I).permute(0, 3, 1, 2)
            else:
                self._views[name] = seg.view(p.shape)

    def update(self):
        if self.optimizer is not None and self.optimizer.shadow is self.shadow:
            return  # already done by FusedAdamW.step()
        p = self.model.flat_parameters()[0]
        ops.axpby(p, self.shadow, 1.0 - self.decay, self.decay, out=self.shadow)

    def apply_shadow(self):
        for name, p in self.model.named_parameters():
            self.backup[name] = p.data
            p.data = self._views[name]

    def restore(self):
        for name, p in self.model.named_parameters():
            p.data = self.backup[name]
        self.backup = {}


# ------------------------------------------------------------------------------------------------
class CyclicCosineDecayLR:
    """scheduler.py:7-119 restated in closed form (the reference class does not construct on torch >= 2.7:
    `verbose=` kwarg, scheduler.py:80).  Stepped once per epoch (train2_seq.py:613-615)."""

    def __init__(self, optimizer, init_decay_epochs=15, min_decay_lr=2.5e-6, restart_interval=10, restart_lr=12.5e-5,
                 warmup_epochs=10, warmup_start_lr=2.5e-6, last_epoch=-1):
        self.optimizer = optimizer
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.a = (init_decay_epochs, min_decay_lr, restart_interval, restart_lr, warmup_epochs, warmup_start_lr)
        self.last_epoch = last_epoch
        self.step()

    @staticmethod
    def _calc(t, T, lr, min_lr):
        return min_lr + (lr - min_lr) * ((1 + math.cos(math.pi * t / T)) / 2)

    def lr_at(self, epoch, base_lr):
        init_decay, min_lr, interval, restart_lr, warm, warm_lr = self.a
        if warm > 0 and epoch < warm:
            return self._calc(epoch, warm, warm_lr, base_lr)
        if epoch < init_decay + warm:
            return self._calc(epoch - warm, init_decay, base_lr, min_lr)
        if interval is None:
            return min_lr
        return self._calc((epoch - init_decay - warm) % interval, interval,
                          base_lr if restart_lr is None else restart_lr, min_lr)

    def step(self):
        self.last_epoch += 1
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = self.lr_at(self.last_epoch, base)

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]


# ------------------------------------------------------------------------------------------------
def compute_acc(y_pred, y_true, top_k=(1, 2, 3)):
    """train2_seq.py:347-360: top-k hit rate (percent, 4 decimals) of argsorted beams."""
    y_pred, y_true = np.asarray(y_pred), np.asarray(y_true)
    if len(y_pred) != len(y_true):
        raise Exception("Number of predicted beams does not match number of labels.")
    hits = [(y_pred[:, :k] == y_true[:, None]).any(axis=1).sum() for k in top_k]
    return np.round(np.asarray(hits, dtype=np.float64) / len(y_true) * 100, 4)


def compute_DBA_score(y_pred, y_true, max_k=3, delta=5):
    """train2_seq.py:363-383: mean over k<=max_k of 1 - mean_i min_{j<=k} min(|pred_ij - y_i|/delta, 1)."""
    y_pred, y_true = np.asarray(y_pred), np.asarray(y_true)
    d = np.minimum(np.abs(y_pred[:, :max_k] - y_true[:, None]) / delta, 1.0)
    yk = [1 - np.minimum.accumulate(d, axis=1)[:, k].mean() for k in range(max_k)]
    return float(np.mean(yk))


def strip_module_prefix(state_dict):
    """Checkpoints written through DataParallel carry a 'module.' prefix (train2_seq.py:276-289, my_test.py:11-22)."""
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}


SCENARIOS = ("scenario31", "scenario32", "scenario33", "scenario34")  # train2_seq.py:196


def per_scenario_metrics(y_pred, y_true, scenario, scenarios=SCENARIOS):
    """train2_seq.py:195-207: top-k accuracy and DBA of every scenario that has samples -> {scenario: (acc, DBA)}."""
    y_pred, y_true, scenario = np.asarray(y_pred), np.asarray(y_true), np.asarray(scenario)
    out = {}
    for s in scenarios:
        sel = scenario == s
        if np.sum(sel) > 0:
            out[s] = (compute_acc(y_pred[sel], y_true[sel]), compute_DBA_score(y_pred[sel], y_true[sel]))
    return out


@torch.no_grad()
def validate(model, batches, ema=None, criterion=None):
    """Engine.validate (train2_seq.py:158-221): optional EMA shadow swap -> eval-mode forward (running BN stats,
    no dropout) -> argsort -> top-k accuracy and DBA, overall and per scenario (:195-207) -> restore.  batches:
    iterable of (fronts, lidars, radars, gps, beamidx[, scenario[, soft_target]]).  Returns (DBA, top-1/2/3 accuracy [%],
    argsorted predictions); with scenario labels the per-scenario table is left in ``validate.last["per_scenario"]``,
    with a criterion and soft targets the mean validation loss (:185-190, 211) in ``validate.last["loss"]``."""
    if ema is not None:
        ema.apply_shadow()
    was_training = model.training
    model.eval()
    preds, truth, scen, losses = [], [], [], []
    for batch in batches:
        fronts, lidars, radars, gps, beamidx = batch[:5]
        logits = model(fronts, lidars, radars, gps)
        preds.append(torch.argsort(logits, dim=1, descending=True).cpu().numpy())
        truth.append(np.asarray(beamidx.cpu() if torch.is_tensor(beamidx) else beamidx))
        if len(batch) > 5 and batch[5] is not None:
            scen.append(np.asarray(batch[5]))
        if criterion is not None:
            tgt = batch[6] if len(batch) > 6 and batch[6] is not None else torch.as_tensor(truth[-1])
            losses.append(float(criterion(logits, tgt.to(logits.device))))
    model.train(was_training)
    if ema is not None:
        ema.restore()
    p, y = np.concatenate(preds), np.concatenate(truth)
    validate.last = dict(per_scenario=per_scenario_metrics(p, y, np.concatenate(scen)) if scen else {},
                         loss=(sum(losses) / len(losses)) if losses else None)
    return compute_DBA_score(p, y), compute_acc(p, y), p


validate.last = {}


@torch.no_grad()
def test(model, batches, target_csv="beam_pred.csv", confidence_csv="beam_pred_confidence_seq.csv"):
    """Engine.test (train2_seq.py:224-253): eval forward -> argsorted beams -> beam_pred.csv, and the softmax
    max-confidence of every sample (:243-246) -> beam_pred_confidence_seq.csv.  batches: iterable of
    (fronts, lidars, radars, gps, ...).  Returns (argsorted predictions, confidences)."""
    was_training = model.training
    model.eval()
    preds, conf = [], []
    for batch in batches:
        logits = model(*batch[:4])
        preds.append(torch.argsort(logits, dim=1, descending=True).cpu().numpy())
        conf.append(torch.softmax(logits, dim=1).max(dim=1)[0].cpu().numpy())
    model.train(was_training)
    p, c = np.concatenate(preds), np.concatenate(conf)
    if target_csv:
        save_pred_to_csv(p, target_csv=target_csv)
    if confidence_csv:
        with open(confidence_csv, "w") as f:  # pandas DataFrame(data=1-D).to_csv layout: unnamed index, column "0"
            f.write(",0\n")
            for i, v in enumerate(c):
                f.write(f"{i},{float(v)!r}\n")
    return p, c


def save_pred_to_csv(y_pred, top_k=(1, 2, 3), target_csv="beam_pred.csv"):
    """train2_seq.py:338-346: 1-based top-k beams, one row per sample (plain csv, no pandas needed)."""
    with open(target_csv, "w") as f:
        f.write("index," + ",".join(f"top-{k} beam" for k in top_k) + "\n")
        for i, row in enumerate(np.asarray(y_pred)):
            f.write(f"{i}," + ",".join(str(int(row[k - 1]) + 1) for k in top_k) + "\n")


class CapturedTrainStep:
    """One whole training iteration (zero_grad -> forward -> focal loss -> backward -> AdamW [+ EMA]) captured into ONE HIP
    graph and replayed: ~2000 kernel launches on 5 streams become one host call.  Everything that changes from step to step
    lives in device memory - the dropout salt (TransFuser._salt, added to every mask counter at run time), the optimizer's
    step count / bias corrections / lr (FusedAdamW._dev), BatchNorm's num_batches_tracked - so a replay is bit-identical to
    the eager iteration it stands for.  step(batch) copies a new batch into the graph's static input buffers first.
    Construction has no training side effect: the warm-up iterations it runs (allocations must exist before a capture)
    are undone, so the first step() is step 1 of the trajectory, as in the reference's Engine.train loop.
    Single process only (the data-parallel all-reduce stays on the eager path)."""

    def __init__(self, model, optimizer, batch, ema=None, warmup=2):
        if model.grad_ready_hook is not None:
            if getattr(optimizer, "_overlap", None) is not None:
                raise RuntimeError("CapturedTrainStep: FusedAdamW.enable_overlap() is active (per-bucket updates issued from "
                                   "the backward walk's gradient hook); capture a step without optimizer overlap")
            raise RuntimeError("CapturedTrainStep: data-parallel training runs eagerly (bucketed all-reduce)")
        self.model, self.optimizer, self.ema = model, optimizer, ema
        dev = model.device
        clone = lambda seq: [t.to(dev, F32).contiguous().clone() for t in seq]  # noqa: E731
        fronts, lidars, radars, gps, target = batch
        self.static = (clone(fronts), clone(lidars), clone(radars), gps.to(dev, F32).contiguous().clone(),
                       target.to(dev, F32).contiguous().clone())
        if warmup < 1:
            raise ValueError("CapturedTrainStep needs warmup >= 1: the lazy one-time allocations (trunk streams, scratch, "
                             "the dropout salt, the bf16 shadow) must exist before the capture")
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        # the warm-up iterations are REAL training steps on the first batch; constructing the stepper must not train, so
        # everything they touch is snapshotted and put back: parameters, both Adam moments, the EMA shadow, BatchNorm
        # running statistics + num_batches_tracked, the optimizer's device scalars / step count and the dropout salt
        with torch.cuda.stream(side):
            snap_t = [model.flat_parameters()[0], optimizer.m, optimizer.v, optimizer._dev, model._nbt]
            if optimizer.shadow is not None:
                snap_t.append(optimizer.shadow)
            if ema is not None and ema.shadow is not None and ema.shadow is not optimizer.shadow:
                snap_t.append(ema.shadow)
            snap_t += [b for n, b in model.named_buffers() if n.endswith(("running_mean", "running_var"))]
            snap = [t.clone() for t in snap_t]
            snap_host = (optimizer.step_count, model._salt_host)
            for _ in range(warmup):   # lazy one-time work (streams, scratch growth, bf16 shadow) outside the capture
                train_iteration(model, optimizer, self.static, ema)
            for t, c in zip(snap_t, snap):
                t.copy_(c)
            optimizer.step_count, model._salt_host = snap_host
            model._salt.fill_(model._salt_host)
            optimizer.zero_grad(set_to_none=True)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        del snap
        # the restore put the device lr slot back to its pre-warm-up value (0 for a fresh optimizer) while the host cache
        # still says "synced": invalidate it, or the captured AdamW would replay with lr 0 until the schedule changes lr
        optimizer._dev_lr = None
        optimizer.sync_lr()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.logits = train_iteration(model, optimizer, self.static, ema)
        # a capture records, it does not execute: the device-side counters did not move, so take the host mirrors back
        optimizer.step_count -= 1
        model._salt_host -= model.SALT_STRIDE
        self.steps_replayed = 0

    def step(self, batch=None):
        if batch is not None:
            fronts, lidars, radars, gps, target = batch
            for dst, src in zip(self.static[0] + self.static[1] + self.static[2] + [self.static[3], self.static[4]],
                                list(fronts) + list(lidars) + list(radars) + [gps, target]):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
        self.optimizer.sync_lr()
        self.graph.replay()
        # host mirrors of the device-side counters the graph advanced
        self.optimizer.step_count += 1
        self.model._note_replayed_step()
        self.steps_replayed += 1
        return self.loss, self.logits

    __call__ = step


def train_iteration(model, optimizer, batch, ema=None, reducer=None):
    """One iteration in the order of Engine.train (train2_seq.py:106-134) on the fused harness path.
    batch = (fronts, lidars, radars, gps, soft_target).  Returns (loss [1], logits)."""
    optimizer.zero_grad(set_to_none=True)
    fronts, lidars, radars, gps, target = batch
    if reducer is not None:
        reducer.begin()
    loss, logits = model.train_step_loss(fronts, lidars, radars, gps, target)
    if reducer is not None:
        reducer.finish()
    optimizer.step()
    if ema is not None:
        ema.update()
    return loss, logits
