"""Data parallelism for the fusion path: one process per GPU, gradients summed with RCCL
(torch.distributed backend "nccl" on ROCm) over xGMI.

The reference uses single-process torch.nn.DataParallel (train2_seq.py:538): per-step parameter
broadcast, input scatter, output gather, gradient reduce onto device 0.  Here each rank owns its
minibatch shard (per-rank BatchNorm statistics, as in DataParallel replicas) and the only exchange
is one gradient sum per step.  Because the gradient arena is laid out in backward-completion order
(model.TransFuser._milestone) the finished gradients are always a contiguous prefix of it, so a
bucket is a plain slice: no packing copy, and each all-reduce is issued the moment its slice is
final, overlapping the rest of the backward walk.  xGMI is point-to-point (7 links x ~153 GB/s per
GPU), so buckets are kept large (>= ~32 MB: ring all-reduce is per-link bandwidth bound, small
messages are latency bound): milestones are coalesced until a bucket reaches `min_bucket_elems`.
The 1/world average is folded into the optimizer kernel (FusedAdamW.grad_scale).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from torchrun-style env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (one-GPU box): DS6G_DIST_BACKEND=gloo and DS6G_FORCE_DEVICE=0 let two ranks share a card;
    # RCCL itself refuses two ranks on one GPU, so production never sets them
    backend = os.environ.get("DS6G_DIST_BACKEND", backend)
    if "DS6G_FORCE_DEVICE" in os.environ:
        local = int(os.environ["DS6G_FORCE_DEVICE"])
    # DS6G_DIST_FORCE_INIT=1: build the process group even for one rank (tests/nccl_worker.py: RCCL on a one-GPU box)
    if (world > 1 or os.environ.get("DS6G_DIST_FORCE_INIT") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class GradReducer:
    """Sums a flat gradient buffer across ranks in contiguous buckets as they become final.

    grad_flat: 1-D tensor (the gradient arena).  `ready(k, lo, hi)` is called by the backward walk
    when grad_flat[lo:hi] is final (milestones arrive in increasing offset order).  `finish()` flushes
    the tail and makes the current stream wait for every outstanding all-reduce."""

    def __init__(self, grad_flat, group=None, min_bucket_elems=8 << 20, flush_at=None):
        self.g = grad_flat
        self.group = group
        self.min_bucket = int(min_bucket_elems)
        # milestone after which whatever has accumulated is sent at once: the bucket that is still open when the backward
        # walk ends cannot overlap anything, so it should hold as little as possible (attach(): everything up to the first
        # GPT stage goes out while layer1 / the stems are still being differentiated; only their ~0.6 M gradients remain)
        self.flush_at = flush_at
        self.works = []
        self.lo = 0
        self.hi = 0
        self.issued = []  # (lo, hi) of every bucket this step, for tests / logging
        # test switches (tests/nccl_worker.py): force_collective runs the all-reduce through the process group even at
        # world size 1 (a one-GPU box can then exercise ProcessGroupNCCL's stream semantics: the collective runs on RCCL's
        # own stream, ordered after the CURRENT stream at the time of the call); probe_stream, when set, takes a copy of
        # every bucket at issue time under exactly that ordering (an event recorded on the current stream), so a test can
        # tell whether the slice was final - i.e. whether every producing side stream had been joined - when it was sent
        self.force_collective = False
        self.probe_stream = None
        self.probes = []
        # subscribers (train.FusedAdamW.enable_overlap): on_begin() at the start of every backward walk, on_bucket(lo, hi,
        # work) right after a bucket was issued - `work` is its all-reduce (None at world size 1): the optimizer updates
        # the bucket's parameters on its own stream as soon as that has completed, while the backward walk goes on
        self.on_begin = None
        self.on_bucket = None

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def begin(self):
        self.works, self.issued, self.probes = [], [], []
        self.lo = self.hi = 0
        if self.on_begin is not None:
            self.on_begin()

    def ready(self, k, lo, hi):
        assert lo == self.hi, "gradient milestones must arrive as a contiguous, growing prefix"
        self.hi = hi
        if self.hi - self.lo >= self.min_bucket or (self.flush_at is not None and k == self.flush_at):
            self._flush()

    def _flush(self):
        if self.hi > self.lo:
            if self.probe_stream is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.probe_stream.wait_event(ev)
                with torch.cuda.stream(self.probe_stream):
                    self.probes.append(self.g[self.lo:self.hi].clone())
            work = None
            if self.world > 1 or (self.force_collective and dist.is_initialized()):
                work = dist.all_reduce(self.g[self.lo:self.hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.works.append(work)
            self.issued.append((self.lo, self.hi))
            if self.on_bucket is not None:
                self.on_bucket(self.lo, self.hi, work)
            self.lo = self.hi

    def finish(self):
        self._flush()
        for w in self.works:
            w.wait()
        self.works = []


def broadcast_parameters(model, src=0, group=None):
    """One-time parameter (and BN buffer) broadcast at start-up; there is no per-step broadcast."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    p, _ = model.flat_parameters()
    dist.broadcast(p, src=src, group=group)
    for b in model.buffers():
        dist.broadcast(b, src=src, group=group)


def attach(model, optimizer, group=None, min_bucket_elems=8 << 20):
    """Wire a GradReducer into the model's backward walk and fold the 1/world average into the optimizer."""
    red = GradReducer(model.flat_parameters()[1], group, min_bucket_elems, flush_at=7)  # 7 = transformer1 (model._milestone)
    model.grad_ready_hook = red.ready  # TransFuser._begin_backward calls red.begin() itself (autograd path included)
    optimizer.grad_scale = 1.0 / red.world
    # independent dropout masks per rank, as DataParallel replicas draw from their own device RNG
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    model.set_dropout_seed(model._base_seed, rank)   # from the UNMIXED seed: attach() twice, or after a resume, mixes once
    return red


def concurrent_side_stream(device, tries=8):
    """A HIP stream whose kernels really run BESIDE those of the calling thread's current stream.  HIP multiplexes streams
    onto a few hardware queues; two streams that share a queue execute in order, which would turn a co-residency rehearsal
    (ds6g_debug_occupy_cus next to the training step) into a serial one.  Probed, not assumed: a 20 ms occupier goes to the
    candidate stream, a tiny kernel to the current one - if the tiny kernel finishes while the occupier is still resident the
    candidate is returned.  -> (stream, kept-alive list of the rejected candidates) or raises."""
    import time

    import torch

    from ._lib import lib
    cur = torch.cuda.current_stream(device)
    probe = torch.zeros(4, device=device)
    rejected = []
    for _ in range(tries):
        cand = torch.cuda.Stream(device)
        torch.cuda.synchronize(device)
        lib().debug_occupy_cus(4, 1024, 20_000, cand.cuda_stream)
        t0 = time.perf_counter()
        probe.add_(1.0)
        cur.synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize(device)
        if dt < 0.010:
            return cand, rejected
        rejected.append(cand)
    raise RuntimeError("no HIP stream found that runs concurrently with the current stream")
