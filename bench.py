#!/usr/bin/env python3
"""Headline benchmark: training samples/s of the fusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
N>1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU, RCCL).  A step = zero_grad -> forward -> sigmoid focal loss (soft target) ->
backward (-> bucketed gradient all-reduce overlapped with backward) -> AdamW step, train mode with
the reference's dropout 0.1 and batch-statistics BatchNorm, on one synthetic batch that is already
resident in HBM.  Workload = BASELINE.json configs[2] (full camera+LiDAR+radar+GPS fusion, 5-step
sequence, bs=12 per GPU); configs[1] ("image-only") is the same compute with zeroed LiDAR/radar
 and configs[0] is the CPU plumbing case timed here as `cpu_baseline`.

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     : dominant kernel = the matrix-core kernel with the largest share of step time; achieved = the MFMA
                 FLOPs its launches EXECUTE / their HIP-event time, measured live in one extra instrumented step
                 (single stream, so brackets are not inflated by the concurrent trunk streams of the timed region);
                 frac = achieved / peak (157.3 TFLOP/s fp32 MFMA, gfx950) - a Winograd kernel executes 2.25x fewer
                 FLOPs than the direct 3x3 conv it replaces, its algorithmic rate is the separate key
                 `algorithmic_tflops`; step_frac = whole-step algorithmic rate / peak; traffic = HBM bytes per launch
                 from the committed PMC passes (profiles/r0N_pmc_traffic.json)
  other_modes  : the same step in the other matrix-core modes, same timing protocol, never the headline: f32x3 / f32x6
                 (split bf16, fp32 storage) and the bf16 configuration of BASELINE configs[1] - "bf16" on the full-fusion
                 batch and "bf16_image_only" on configs[1]'s zeroed LiDAR / radar batch - each bf16 entry with step_frac
                 against the 2.5 PFLOP/s dense bf16 peak and the roofline of its dominant bgemm_kernel instantiation
  dba          : the second half of BASELINE's metric ("DBA top-k on held-out", train2_seq.py:363-383): a fresh model
                 trained for 150 steps on the synthetic learnable beam task on the HIP path (tools/train_synthetic.py's
                 protocol), held-out DBA / top-k by train.validate at steps 0 / 50 / 100 / 150, the committed CPU-oracle
                 curve of the same protocol beside it (profiles/r02_train_synthetic_dba_cpu_oracle_150.jsonl)
  cpu_baseline : the CPU oracle (oracle/, torch fp32 on the host cores): bs=2 (3 warm-up + 10 timed) and bs=12
                 (1 warm-up + 3 timed) steps of fwd + focal + bwd + AdamW, median step time, CPU model and threads.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (no 2:1 sparsity)
VARIANT_NAMES = {0: "128x128", 1: "128x64", 2: "64x64"}
MODE_NAMES = {0: "fwd", 1: "dgrad", 2: "wgrad"}


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = n
    try:  # cgroup v2 cpu quota (the GPU box gives each job a share of a 256-thread host)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:
        pass
    return max(1, min(n, quota))


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(batch=12, steps=3, warmup=1, small_batch=2, small_steps=10, small_warmup=3):
    """fwd + focal + bwd + AdamW of the CPU oracle on the host cores (reference CPU path restated), SURVEY 8(d)
    protocol: all threads of the job's cgroup share, fp32, median step time at bs=2 (BASELINE configs[0]) and at the
    benchmark's bs=12.  `value` = the bs=12 rate (the same workload the GPU line measures)."""
    import statistics

    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    cores = host_threads()
    torch.set_num_threads(cores)
    cfg = fr.RefConfig()

    def run(bs, n_warm, n_timed):
        sd = fr.make_state(cfg, seed=0, scheme="init")
        params = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not fr.is_buffer(k)]
        opt = torch.optim.AdamW(params, lr=1e-4)
        imgs, lids, rads, gps, target, _ = fr.make_inputs(cfg, bs, seed=100)
        times = []
        for it in range(n_warm + n_timed):
            t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            logits = fr.transfuser_forward(sd, imgs, lids, rads, gps, cfg, fr.Ctx(training=True, dropout=True))
            loss = tr.sigmoid_focal_loss(logits, target)
            loss.backward()
            opt.step()
            times.append(time.perf_counter() - t0)
        timed = times[n_warm:]
        return bs / statistics.median(timed), timed

    small, small_t = run(small_batch, small_warmup, small_steps) if small_steps > 0 else (None, [])
    big, big_t = run(batch, warmup, steps)
    return dict(value=big, unit="samples/s", cores=cores, kind="port", cpu_model=cpu_model_name(),
                sample=f"median of {len(big_t)} timed steps (+{warmup} warm-up) of bs={batch}; bs={small_batch}: median of "
                       f"{len(small_t)} (+{small_warmup} warm-up); fp32, torch {torch.__version__} CPU, "
                       f"oracle/fusion_ref.py fwd+focal+bwd+AdamW, {cores} threads",
                by_batch={str(batch): dict(samples_per_s=big, step_s=[round(t, 3) for t in big_t]),
                          **({str(small_batch): dict(samples_per_s=small, step_s=[round(t, 3) for t in small_t])}
                             if small is not None else {})})


def variant_name(v, dtype="f32", short=False):
    """variant code (include/ds6g.h): 10000 * wide + 1000 * epilogue + 100 * uniform-walk + 10 * mode + tile  ->  the
    template instantiation exactly as rocprofv3 names it: igemm_kernel<mode, BM, BN, epilogue, BK, bf16, walk>"""
    if v >= 30000:  # bgemm.hip (bf16-stored operands): 30000 + 100 * bf16-output + 10 * mode + tile (0 = 128x128, 1 = 64x64)
        o16, mode, tile = (v - 30000) // 100, (v - 30000) % 100 // 10, v % 10
        t = ("128x128", "64x64")[tile]
        return (f"bf16-stored/{MODE_NAMES[mode]}/{t}" + ("/bf16-out" if o16 else "")) if short else \
            f"bgemm_kernel<{mode}, {t.split('x')[0]}, {t.split('x')[1]}, {o16}, *>"
    if v >= 20000:  # winograd.hip (flops recorded = those of the direct 3x3 conv it replaces)
        return ("winograd_fwd_kernel<1>", "winograd_wgrad_kernel", "winograd_pc_kernel<0>")[v - 20000] if not short else \
            ("winograd/fwd+dgrad", "winograd/wgrad", "winograd/fwd+dgrad (producer-consumer)")[v - 20000]
    bfm = ("f32", "bf16", "f32x3", "f32x6").index(dtype)
    wide, epi, walk, mode, tile = v // 10000, v % 10000 // 1000, v % 1000 // 100, v % 100 // 10, v % 10
    bm, bn = VARIANT_NAMES[tile].split("x")
    if short:
        return (f"{MODE_NAMES[mode]}/{bm}x{bn}" + ("/k32" if wide else "") + ("/epi" if epi else "") +
                ("" if walk else "/general-walk"))
    return f"igemm_kernel<{mode}, {bm}, {bn}, {epi}, {32 if wide else 16}, {bfm}, {walk}>"


def peak_tflops_for(dtype):
    # f32x3 executes three bf16 MFMA flops per algorithmic flop: its ceiling in algorithmic flops is a third of the bf16 peak
    return {"f32": PEAK_FP32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "f32x3": PEAK_BF16_MFMA_TFLOPS / 3,
            "f32x6": PEAK_BF16_MFMA_TFLOPS / 6}[dtype]


WINOGRAD_EXEC = 1.0 / 2.25   # F(2x2,3x3): 16 multiplies per 2x2 output tile and channel pair instead of 36


def roofline_from_records(agg, dtype, traffic):
    """agg: {variant: [launches, algorithmic flops, ms]} of ONE instrumented step.  The roofline kernel is the matrix-core
    kernel with the largest share of step time.  `achieved` / `frac` are the MFMA FLOPs the kernel EXECUTES per second
    (a fraction of the matrix peak, <= 1 by construction); a Winograd kernel's rate in direct-conv FLOPs (SURVEY 8d counts
    a 3x3 conv as 9 taps) is reported separately as `algorithmic_tflops`."""
    peak = peak_tflops_for(dtype)
    dom = max(agg, key=lambda v: agg[v][2])

    def one(v):
        cnt, fl, ms = agg[v]
        ex = WINOGRAD_EXEC if 20000 <= v < 30000 else 1.0
        algo = fl / (ms * 1e-3) / 1e12
        return dict(kernel=variant_name(v, dtype), achieved=algo * ex, peak=peak, unit="TFLOP/s", frac=algo * ex / peak,
                    traffic=traffic(v) if dtype == "f32" else None, algorithmic_tflops=algo, launches_per_step=cnt,
                    avg_launch_us=ms * 1e3 / cnt, flops_per_launch=fl * ex / cnt, algorithmic_flops_per_launch=fl / cnt)

    roof = dict(bound="mfma", **one(dom))
    if 20000 <= dom < 30000:
        roof["note"] = ("Winograd F(2x2,3x3): achieved / frac count the MFMA flops the kernel executes (2.25x fewer than the "
                        "direct 3x3 conv); algorithmic_tflops is the direct-conv rate, see DESIGN.md 3.1b")
        direct = [v for v in agg if not 20000 <= v < 30000]
        if direct:
            d = one(max(direct, key=lambda v: agg[v][2]))
            d.pop("peak"), d.pop("unit")
            roof["largest_direct_kernel"] = d
    tot_ms = sum(v[2] for v in agg.values())
    tot_fl = sum(v[1] for v in agg.values())
    tot_ex = sum(v[1] * (WINOGRAD_EXEC if 20000 <= k < 30000 else 1.0) for k, v in agg.items())
    roof["igemm_family"] = dict(achieved=tot_ex / (tot_ms * 1e-3) / 1e12, algorithmic_tflops=tot_fl / (tot_ms * 1e-3) / 1e12,
                                ms_per_step=tot_ms, flops_per_step=tot_fl,
                                by_variant={variant_name(v, dtype, short=True):
                                            dict(launches=a[0], ms=round(a[2], 3),
                                                 tflops=round(a[1] * (WINOGRAD_EXEC if 20000 <= v < 30000 else 1.0) / (a[2] * 1e-3) / 1e12, 2))
                                            for v, a in sorted(agg.items())})
    return roof


def pmc_traffic_table():
    """HBM bytes per launch by kernel name from the committed rocprofv3 PMC passes (profiles/): FETCH_SIZE and WRITE_SIZE
    are collected in separate runs of this same command, so bench.py cannot measure them live; {} when absent."""
    for rnd in ("r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
        if os.path.exists(path):
            return {k: v["hbm_bytes_per_launch"] for k, v in json.load(open(path))["kernels"].items()}
    return {}


class IgemmTimer:
    """Live per-kernel timing of one step: the library brackets every implicit-GEMM KERNEL launch with HIP events on
    its launch stream (ds6g_profile_begin/end), so the durations are the kernels' own (the split-K reduction that
    follows a wgrad is not included) and line up with the rocprofv3 --kernel-trace averages under profiles/."""

    CAP = 1 << 14

    def install(self):
        from deepsense6g_tii_amd._lib import lib
        lib().profile_begin(self.CAP)

    def summary(self):
        import ctypes
        from deepsense6g_tii_amd._lib import lib
        var = (ctypes.c_int * self.CAP)()
        fl = (ctypes.c_double * self.CAP)()
        ms = (ctypes.c_float * self.CAP)()
        n = lib().profile_end(var, fl, ms, self.CAP)
        agg = {}
        for i in range(n):
            d = agg.setdefault(var[i], [0, 0.0, 0.0])
            d[0] += 1
            d[1] += fl[i]
            d[2] += ms[i]
        return agg


PEAK_HBM_TBS = 8.0   # MI355X_MICROARCH.md: HBM3E ~8 TB/s spec (~6.3 achievable by a streaming kernel)


def hbm_families(model, opt, batch, ema, reducer):
    """The HBM-bound half of the roofline: one extra single-stream training step (outside the timed region) in which every call
    of the BatchNorm / LayerNorm / AdamW entry points is bracketed by HIP events on its launch stream; per family: calls,
    ms per step, ALGORITHMIC bytes (the tensors the operation must read and write once per pass, DESIGN.md section 3: BN
    statistics 1 read; BN apply 1 read (+ residual) + 1 write; BN backward two passes over dy and x (+ the mask tensor) and
    one write of dx (+ dres); LayerNorm forward 1 + 1, backward reads dy, x (+ add) and writes dx (+ its dropout copy);
    AdamW 28 B per parameter, + 12 with the EMA shadow) and achieved TB/s = bytes / time.  A call = its reduction, finalize
    and apply kernels together, so small layers read low (kernel-boundary latency), large ones approach the streaming rate."""
    from deepsense6g_tii_amd._lib import lib
    L = lib()
    recs = []
    saved = {}

    def wrap(name, fam, nbytes):
        orig = getattr(L, name)
        saved[name] = orig

        def call(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(*a)
            e1.record()
            recs.append((fam, float(nbytes(a)), e0, e1))
        setattr(L, name, call)

    n_ = lambda v: 1 if v else 0  # noqa: E731
    for pre, eb in (("", 4), ("bf16_", 2)):   # fp32 storage, and the bf16-storage variants (other_modes)
        wrap(pre + "bn_stats", "batchnorm statistics", lambda a, eb=eb: a[1] * a[2] * eb)
        wrap(pre + "bn_apply", "batchnorm apply (+ReLU, +residual)", lambda a, eb=eb: (2 + n_(a[5])) * a[7] * a[8] * eb)
        wrap(pre + "bn_bwd", "batchnorm backward (reduce + apply)",
             lambda a, eb=eb: (2 * (2 + n_(a[1])) + 1 + n_(a[10])) * a[11] * a[12] * eb)
    wrap("layernorm_fwd", "layernorm forward", lambda a: 2 * a[6] * a[7] * 4)
    wrap("layernorm_bwd", "layernorm backward", lambda a: (3 + n_(a[5]) + n_(a[12])) * a[9] * a[10] * 4)
    wrap("adamw_step_dev", "adamw (+ema)", lambda a: a[5] * (28 + (12 if a[4] else 0)))
    ms_flag = model.multi_stream
    model.multi_stream = False
    try:
        from deepsense6g_tii_amd.train import train_iteration
        train_iteration(model, opt, batch, ema, reducer)
        torch.cuda.synchronize()
    finally:
        model.multi_stream = ms_flag
        for name, orig in saved.items():
            setattr(L, name, orig)
    fam = {}
    for f, nb, e0, e1 in recs:
        d = fam.setdefault(f, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += nb
        d[2] += e0.elapsed_time(e1)
    return {f: dict(calls_per_step=c, ms_per_step=round(ms, 3), algorithmic_gb_per_step=round(nb / 1e9, 3),
                    achieved_tb_s=round(nb / (ms * 1e-3) / 1e12, 2), frac_of_hbm_peak=round(nb / (ms * 1e-3) / 1e12 / PEAK_HBM_TBS, 3))
            for f, (c, nb, ms) in sorted(fam.items(), key=lambda kv: -kv[1][2]) if ms > 0}


def eval_leg(model, batch, nbatch, steps=20):
    """f2 in the driver line: eval-mode inference (BatchNorm folded into the conv weights, no dropout, no tape) on the timed
    configuration's batch, samples/s (train2_seq.py:158-221 forward part)."""
    was = model.training
    model.eval()
    try:
        with torch.no_grad():
            for _ in range(3):
                model(*batch[:4])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = model(*batch[:4])
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        return dict(value=nbatch * steps / el, unit="samples/s", ms_per_batch=el / steps * 1e3, batch=nbatch, steps=steps,
                    what="eval-mode forward, BatchNorm folded into the conv weights (ds6g_bn_fold), fp32, "
                         "(B, 64) logits; f2 of SURVEY 8", finite=bool(torch.isfinite(out).all()))
    finally:
        model.train(was)


def seq10_leg(dev, batch=4, steps=5):
    """f4 in the driver line: one training configuration of the 30 -> 5 variant (/root/reference/model2_seq_30to5.py: seq_len
    10 => 1922 tokens per sample, GRU beam-sequence head, pred_len 5, gradient clip 3.0 of train2_seq_30to5.py:120), bs=4,
    exact fp32, dropout 0.1: ms per step and samples/s."""
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser30to5
    from deepsense6g_tii_amd.synthetic import make_batch
    from deepsense6g_tii_amd.train import FusedAdamW, train_iteration
    cfg = GlobalConfig(seq_len=10, pred_len=5)
    torch.manual_seed(100)
    model = TransFuser30to5(cfg, dev)
    model.train()
    opt = FusedAdamW(model, lr=1e-4, max_grad_norm=3.0)
    fronts, lidars, radars, gps, target, _ = make_batch(batch, cfg.seq_len, cfg.n_views, cfg.add_velocity, seed=100, device=dev)
    tgt = target[:, None, :].expand(batch, cfg.pred_len, 64).contiguous()
    b = (fronts, lidars, radars, gps, tgt)
    for _ in range(2):
        train_iteration(model, opt, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = train_iteration(model, opt, b)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = dict(value=batch * steps / el, unit="samples/s", ms_per_step=el / steps * 1e3, batch=batch, steps=steps, seq_len=10,
               tokens=cfg_tokens(cfg), loss=float(loss),
               what="TransFuser30to5 training step (seq_len 10, 1922 tokens, GRU head, pred_len 5, global-norm clip 3.0), fp32")
    del model, opt
    torch.cuda.empty_cache()
    return out


def cfg_tokens(cfg):
    return (cfg.n_views + 2) * cfg.seq_len * cfg.vert_anchors * cfg.horz_anchors + 2


def dba_leg(dev, steps=150, batch=12, pool=16, eval_batches=4, eval_every=50):
    """Held-out DBA / top-k of the HIP path on the synthetic learnable task: tools/train_synthetic.py's protocol (fresh
    model from torch.manual_seed(100), training batches of seeds 100 + i, held-out seeds 10 000 + i, AdamW lr 1e-4, focal
    loss on the soft target, dropout 0.1, train-mode BN; evaluation = train.validate: eval-mode forward -> argsort ->
    compute_DBA_score / compute_acc, train2_seq.py:158-221, 347-383).  The CPU-oracle run of the same protocol (same init,
    same batches, torch's own dropout stream; oracle/train_synthetic_cpu.py, 2 h on 7 threads) is read from profiles/."""
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from deepsense6g_tii_amd.synthetic import make_batch
    from deepsense6g_tii_amd.train import FusedAdamW, train_iteration, validate
    t_start = time.perf_counter()
    torch.manual_seed(100)
    cfg = GlobalConfig()
    model = TransFuser(cfg, dev)
    opt = FusedAdamW(model, lr=1e-4)
    held_out = []
    for i in range(eval_batches):  # disjoint seed range
        f, l, r, g, _, beam = make_batch(batch, seed=10_000 + i, device=dev, learnable=True)
        held_out.append((f, l, r, g, beam))
    train_pool = [make_batch(batch, seed=100 + i, device=dev, learnable=True)[:5] for i in range(pool)]

    def evaluate(step, loss=None):
        dba, acc, _ = validate(model, held_out)
        model.train()
        return dict(step=step, dba=dba, top123=acc.tolist(), **({} if loss is None else {"loss": float(loss)}))

    curve = [evaluate(0)]
    model.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(1, steps + 1):
        loss, _ = train_iteration(model, opt, train_pool[step % pool])
        if step % eval_every == 0 or step == steps:
            curve.append(evaluate(step, loss))
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    cpu_curve = None
    path = os.path.join(ROOT, "profiles", "r02_train_synthetic_dba_cpu_oracle_150.jsonl")
    if os.path.exists(path):
        rows = [json.loads(line) for line in open(path) if line.strip()]
        cpu_curve = [dict(step=r["step"], dba=r["dba"], top123=r["top123"]) for r in rows]
    at = {r["step"]: r["dba"] for r in (cpu_curve or [])}
    del model, opt, train_pool, held_out
    torch.cuda.empty_cache()
    return dict(metric="DBA (mean over k <= 3 of 1 - mean_i min_j<=k min(|pred_ij - y_i| / 5, 1)) and top-1/2/3 accuracy [%] "
                       "on held-out synthetic sequences", value=curve[-1]["dba"], top123=curve[-1]["top123"],
                steps=steps, held_out_samples=eval_batches * batch, train_batches=pool, batch=batch,
                curve=curve, cpu_oracle_curve=cpu_curve, cpu_oracle_dba_at_same_step=at.get(steps),
                cpu_oracle_source="profiles/r02_train_synthetic_dba_cpu_oracle_150.jsonl (oracle/train_synthetic_cpu.py: same "
                                  "init / batches / protocol on the CPU oracle, torch's dropout stream)",
                dtype="f32", data="synthetic learnable task (deepsense6g_tii_amd/synthetic.py: the beam index painted into "
                                  "every modality and the GPS angle); no dataset offline",
                train_seconds=train_s, total_seconds=time.perf_counter() - t_start)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=12, help="per-GPU batch (sequences)")
    ap.add_argument("--ema", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-modes", action="store_true",
                    help="skip the extra f32x3 / f32x6 / bf16 / bf16-image-only timings beside the exact-fp32 value")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the hbm_families / eval / seq10 legs")
    ap.add_argument("--no-dba", action="store_true", help="skip the held-out DBA leg (150 training steps on the synthetic learnable task)")
    ap.add_argument("--dba-steps", type=int, default=150)
    ap.add_argument("--dtype", choices=("f32", "bf16", "f32x3", "f32x6"), default="f32",
                    help="matrix-core mode: f32 = exact fp32 MFMA (parity path, default); bf16 = operands rounded to bf16, "
                         "fp32 accumulate/storage (BASELINE configs[1]-style throughput configuration); f32x3 = split bf16 "
                         "(hi*hi + hi*lo + lo*hi, ~2^-16 relative product error, fp32 accumulate/storage)")
    ap.add_argument("--image-only", action="store_true",
                    help="BASELINE configs[1]: zeroed LiDAR / radar inputs - the reference's own 'zerolike' missing-modality "
                         "semantics (mambafuser_seq.py:384-391); same kernels and FLOPs, other input statistics")
    ap.add_argument("--graph", type=int, default=0,
                    help="1 (single process): the timed steps replay ONE captured HIP graph of the whole iteration "
                         "(train.CapturedTrainStep, bit-identical to the eager iteration); 0: eager launches.  Data-parallel "
                         "runs (--gpus > 1) are always eager (bucketed all-reduce issued from the backward walk)")
    ap.add_argument("--overlap-optimizer", type=int, default=0,
                    help="1: the AdamW (+EMA) update of every gradient bucket runs on a side stream as soon as the bucket is "
                         "final (under data parallelism: all-reduced), overlapping the rest of the backward walk "
                         "(FusedAdamW.enable_overlap; bit-identical to the plain order; measured +0.1 % at N = 1: the 0.4 ms pass "
                         "contends for HBM with the kernels it runs beside); 0 (default): one update after backward")
    ap.add_argument("--single-stream", action="store_true",
                    help="run the three trunks on one stream (per-kernel profiling: durations are not overlapped)")
    ap.add_argument("--debug-flags", type=lambda v: int(v, 0), default=0, help="ds6g_set_debug_flags (tuning experiments)")
    ap.add_argument("--cpu-batch", type=int, default=12)
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-oracle steps at --cpu-batch (after 1 warm-up)")
    ap.add_argument("--cpu-small-steps", type=int, default=10, help="timed CPU-oracle steps at bs=2 (after 3 warm-up); 0 = skip")
    args = ap.parse_args()

    from deepsense6g_tii_amd import dist as ddist
    from deepsense6g_tii_amd import ops
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from deepsense6g_tii_amd.synthetic import make_batch
    from deepsense6g_tii_amd.train import EMA, CapturedTrainStep, FusedAdamW, train_iteration

    # the process group first: under torch.distributed.run nothing may touch the GPU before init_process_group
    rank, world, local = ddist.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ops.set_compute_mode(args.dtype)
    if args.debug_flags:
        from deepsense6g_tii_amd._lib import lib
        lib().set_debug_flags(args.debug_flags)
    peak_tflops = peak_tflops_for(args.dtype)

    cfg = GlobalConfig()
    torch.manual_seed(100)  # reference seeds everything with 100 (train2_seq.py:430-437)
    model = TransFuser(cfg, dev)
    model.train()
    model.multi_stream = not args.single_stream
    opt = FusedAdamW(model, lr=1e-4, ema_decay=0.999 if args.ema else None)
    ema = None
    if args.ema:
        ema = EMA(model, 0.999, opt)
        ema.register()
    reducer = None
    if world > 1:
        ddist.broadcast_parameters(model)
        reducer = ddist.attach(model, opt)
    if args.overlap_optimizer and not args.graph:
        opt.enable_overlap(model)
    fronts, lidars, radars, gps, target, _ = make_batch(args.batch, cfg.seq_len, cfg.n_views, cfg.add_velocity,
                                                        seed=100 + rank, device=dev)
    if args.image_only:
        lidars = [torch.zeros_like(t) for t in lidars]
        radars = [torch.zeros_like(t) for t in radars]
    batch = (fronts, lidars, radars, gps, target)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    use_graph = bool(args.graph) and world == 1
    if use_graph:
        stepper = CapturedTrainStep(model, opt, batch, ema, warmup=2)
        run_step = stepper.step
    else:
        run_step = lambda: train_iteration(model, opt, batch, ema, reducer)  # noqa: E731
    for _ in range(args.warmup):
        loss, _ = run_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = run_step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())

    # ---- roofline of the dominant kernel: one extra instrumented step (not part of the timed region) ----
    roof = None
    timer = IgemmTimer()
    timer.install()
    ms_flag = model.multi_stream
    model.multi_stream = False  # kernels of concurrent trunk streams would inflate each other's event brackets
    try:
        train_iteration(model, opt, batch, ema, reducer)
        agg = timer.summary()
    finally:
        model.multi_stream = ms_flag
    if agg:
        table = pmc_traffic_table()
        roof = roofline_from_records(agg, args.dtype, lambda v: table.get(variant_name(v, args.dtype)))

    # ---- the HBM-bound kernel families of the same step, the eval-mode forward (f2) and the seq_len-10 variant (f4):
    # separate untimed legs of a few seconds each
    hbm = ev = None
    if rank == 0 and world == 1 and not args.no_extra_legs:
        try:
            hbm = hbm_families(model, opt, batch, ema, reducer)
        except Exception as e:
            hbm = {"error": repr(e)}
        try:
            ev = eval_leg(model, batch, args.batch)
        except Exception as e:
            ev = {"error": repr(e)}

    # ---- the other matrix-core modes next to the exact one (same model / timing protocol; reported beside `value`, never
    # as it): split-bf16 on the same batch, and the bf16 configuration BASELINE configs[1] names ----
    alt = None
    if args.dtype == "f32" and not args.no_alt_modes:
        what = {"f32x3": "split-bf16 products (hi*hi + hi*lo + lo*hi, ~2^-16), fp32 accumulate and storage; "
                         "ds6g_set_compute_mode(2); Winograd off (direct implicit GEMM); tests/test_bf16_gpu.py",
                "f32x6": "three-way bf16 split, six products (fp32-grade, ~2^-23) in the direct kernels and attention, fp32 "
                         "Winograd kept for the 3x3/1 convs; ds6g_set_compute_mode(3); tests/test_bf16_gpu.py",
                "bf16": "BASELINE configs[1] / [4] arithmetic on the full-fusion batch of configs[2]: bf16 forward / backward "
                        "(activations, their gradients and a per-step weight shadow stored as bf16, csrc/bgemm.hip; fp32 master "
                        "weights, statistics, accumulators, loss, AdamW); ds6g_set_compute_mode(1); parity unpinned (the "
                        "reference has no reduced-precision path), builder-declared bars in tests/test_bf16_gpu.py",
                "bf16_image_only": "BASELINE configs[1] literally: image-only (LiDAR / radar inputs zeroed, the reference's "
                                   "'zerolike' missing-modality semantics), bs=12, bf16 forward / backward"}
        zeroed = (fronts, [torch.zeros_like(t) for t in lidars], [torch.zeros_like(t) for t in radars], gps, target)
        alt = {}
        for mode in ("f32x3", "f32x6", "bf16", "bf16_image_only"):
            cmode = "bf16" if mode.startswith("bf16") else mode
            mbatch = zeroed if mode == "bf16_image_only" else batch
            try:
                with torch.no_grad():
                    model.eval()
                    ops.set_compute_mode("f32")
                    ref_logits = model(*mbatch[:4]).float().clone()
                    ops.set_compute_mode(cmode)
                    m_logits = model(*mbatch[:4]).float()
                    dev_rel = float((m_logits - ref_logits).abs().max() / ref_logits.abs().max())
                    model.train()
                for _ in range(2):
                    train_iteration(model, opt, mbatch, ema, reducer)
                sync()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    train_iteration(model, opt, mbatch, ema, reducer)
                sync()
                el = time.perf_counter() - t0
                if world > 1:
                    t = torch.tensor([el], device=dev, dtype=torch.float64)
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    el = float(t.item())
                v = args.batch * world * args.steps / el
                alt[mode] = {"value": v, "unit": "samples/s", "ms_per_step": el / args.steps * 1e3, "dtype": cmode,
                             "step_frac": v / world * 559.3e9 / 1e12 / peak_tflops_for(cmode),
                             "eval_logits_max_dev_vs_f32_rel": dev_rel, "what": what[mode]}
                if cmode == "bf16":   # roofline of its dominant kernel: one instrumented single-stream step in this mode
                    t2 = IgemmTimer()
                    t2.install()
                    model.multi_stream = False
                    try:
                        train_iteration(model, opt, mbatch, ema, reducer)
                        agg2 = t2.summary()
                    finally:
                        model.multi_stream = ms_flag
                    bg = {k: a for k, a in agg2.items() if k >= 30000}     # the bf16-stored GEMM family (the stems stay igemm)
                    if bg:
                        r2 = roofline_from_records(bg, "bf16", lambda v: None)
                        r2["family"] = r2.pop("igemm_family")
                        alt[mode]["roofline"] = r2
            except Exception as e:  # an extra mode never takes the headline line down with it
                alt[mode] = {"error": repr(e)}
            finally:
                ops.set_compute_mode("f32")
                model.train()

    # ---- the second half of the metric: held-out DBA / top-k after a short training run on the HIP path ----
    dba = None
    if rank == 0 and world == 1 and not args.no_dba and args.dtype == "f32":
        try:
            dba = dba_leg(dev, steps=args.dba_steps)
        except Exception as e:
            dba = {"error": repr(e)}

    seq10 = None
    if rank == 0 and world == 1 and not args.no_extra_legs and args.dtype == "f32":
        del model, opt
        torch.cuda.empty_cache()
        try:
            seq10 = seq10_leg(dev)
        except Exception as e:
            seq10 = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_batch, args.cpu_steps, 1, 2, args.cpu_small_steps, 3)

    if rank == 0:
        samples = args.batch * world * args.steps
        value = samples / elapsed
        out = {
            "metric": f"training samples/sec (5-frame seq, bs={args.batch} per GPU)",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: image-only (LiDAR / radar inputs zeroed), " if args.image_only else
                                    "BASELINE configs[2]: full camera+LiDAR+radar+GPS fusion, ") +
                                   f"5x(3+1+2)x256x256 + 2x2 GPS per sample, bs={args.batch} per GPU, sigmoid focal loss, "
                                   "AdamW, train-mode BN, dropout 0.1",
                       "global_batch": args.batch * world, "seq_len": cfg.seq_len,
                       "parallelism": f"dp{world}", "ema": bool(args.ema),
                       "launch": "one HIP graph per step" if use_graph else "eager",
                       "optimizer": "AdamW per gradient bucket on a side stream during backward" if (args.overlap_optimizer and not args.graph)
                                    else "AdamW after backward"},
            "loss": final_loss,
            "algorithmic_gflop_per_sample": 559.3,
            "model_tflops": value * 559.3e9 / 1e12,
            "step_frac": value / world * 559.3e9 / 1e12 / peak_tflops,   # per-GPU algorithmic rate / matrix peak of the dtype
            "roofline": roof,
            "cpu_baseline": cpu,
            "other_modes": alt,
            "dba": dba,
            "hbm_families": hbm,
            "eval": ev,
            "seq10": seq10,
        }
        if cpu:
            out["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
