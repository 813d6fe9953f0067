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
  roofline     : dominant kernel = the implicit-GEMM instantiation with the largest share of step time;
                 achieved = algorithmic FLOPs of its launches / their HIP-event time, measured live in one
                 extra instrumented step (single stream, so brackets are not inflated by the concurrent trunk
                 streams of the timed region); peak = 157.3 TFLOP/s (fp32 MFMA, gfx950); traffic = HBM bytes per
                 launch from the committed PMC passes (profiles/r01_pmc_traffic.json)
  cpu_baseline : the CPU oracle (oracle/, torch fp32 on the host cores) timed on a bounded sample.
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


def cpu_baseline(batch, steps):
    """fwd + focal + bwd + AdamW of the CPU oracle on the host cores (reference CPU path restated)."""
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    cores = host_threads()
    torch.set_num_threads(cores)
    cfg = fr.RefConfig()
    sd = fr.make_state(cfg, seed=0, scheme="init")
    params = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not fr.is_buffer(k)]
    opt = torch.optim.AdamW(params, lr=1e-4)
    imgs, lids, rads, gps, target, _ = fr.make_inputs(cfg, batch, seed=100)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        logits = fr.transfuser_forward(sd, imgs, lids, rads, gps, cfg, fr.Ctx(training=True, dropout=True))
        loss = tr.sigmoid_focal_loss(logits, target)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    timed = times[1:]
    return dict(value=batch * len(timed) / sum(timed), unit="samples/s", cores=cores, kind="port",
                sample=f"{len(timed)} timed steps (+1 warm-up) of bs={batch}, fp32, torch {torch.__version__} CPU, "
                       f"oracle/fusion_ref.py fwd+focal+bwd+AdamW")


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

    def uninstall(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=12, help="per-GPU batch (sequences)")
    ap.add_argument("--ema", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-modes", action="store_true", help="skip the extra f32x3 / f32x6 timings beside the exact-fp32 value")
    ap.add_argument("--dtype", choices=("f32", "bf16", "f32x3", "f32x6"), default="f32",
                    help="matrix-core mode: f32 = exact fp32 MFMA (parity path, default); bf16 = operands rounded to bf16, "
                         "fp32 accumulate/storage (BASELINE configs[1]-style throughput configuration); f32x3 = split bf16 "
                         "(hi*hi + hi*lo + lo*hi, ~2^-16 relative product error, fp32 accumulate/storage)")
    ap.add_argument("--image-only", action="store_true",
                    help="BASELINE configs[1]: zeroed LiDAR / radar inputs - the reference's own 'zerolike' missing-modality "
                         "semantics (mambafuser_seq.py:384-391); same kernels and FLOPs, other input statistics")
    ap.add_argument("--single-stream", action="store_true",
                    help="run the three trunks on one stream (per-kernel profiling: durations are not overlapped)")
    ap.add_argument("--debug-flags", type=lambda v: int(v, 0), default=0, help="ds6g_set_debug_flags (tuning experiments)")
    ap.add_argument("--cpu-batch", type=int, default=12)
    ap.add_argument("--cpu-steps", type=int, default=2)
    args = ap.parse_args()

    from deepsense6g_tii_amd import dist as ddist
    from deepsense6g_tii_amd import ops
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from deepsense6g_tii_amd.synthetic import make_batch
    from deepsense6g_tii_amd.train import EMA, FusedAdamW, train_iteration

    ops.set_compute_mode(args.dtype)
    if args.debug_flags:
        from deepsense6g_tii_amd._lib import lib
        lib().set_debug_flags(args.debug_flags)
    # f32x3 executes three bf16 MFMA flops per algorithmic flop: its ceiling in algorithmic flops is a third of the bf16 peak
    peak_tflops = {"f32": PEAK_FP32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "f32x3": PEAK_BF16_MFMA_TFLOPS / 3, "f32x6": PEAK_BF16_MFMA_TFLOPS / 6}[args.dtype]
    rank, world, local = ddist.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

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

    for _ in range(args.warmup):
        loss, _ = train_iteration(model, opt, batch, ema, reducer)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = train_iteration(model, opt, batch, ema, reducer)
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
        timer.uninstall()
        model.multi_stream = ms_flag
    def pmc_traffic(variant):
        """HBM bytes per launch of an igemm instantiation from the committed rocprofv3 PMC passes (profiles/):
        FETCH_SIZE and WRITE_SIZE are collected in separate runs of this same command, so bench.py cannot measure
        them live; null when the summary is absent."""
        path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if not os.path.exists(path):
            return None
        ks = json.load(open(path))["kernels"]
        v = ks.get(vname(variant))
        return v["hbm_bytes_per_launch"] if v else None

    def vname(v, short=False):
        # variant code (include/ds6g.h): 10000 * wide + 1000 * epilogue + 100 * uniform-walk + 10 * mode + tile  ->  the template
        # instantiation exactly as rocprofv3 names it: igemm_kernel<mode, BM, BN, epilogue, BK, bf16, walk>
        if v >= 20000:  # winograd.hip (flops recorded = those of the direct 3x3 conv it replaces)
            return ("winograd_fwd_kernel<1>", "winograd_wgrad_kernel")[v - 20000] if not short else \
                ("winograd/fwd+dgrad", "winograd/wgrad")[v - 20000]
        bfm = ("f32", "bf16", "f32x3", "f32x6").index(args.dtype)
        wide, epi, walk, mode, tile = v // 10000, v % 10000 // 1000, v % 1000 // 100, v % 100 // 10, v % 10
        bm, bn = VARIANT_NAMES[tile].split("x")
        if short:
            return (f"{MODE_NAMES[mode]}/{bm}x{bn}" + ("/k32" if wide else "") + ("/epi" if epi else "") +
                    ("" if walk else "/general-walk"))
        return f"igemm_kernel<{mode}, {bm}, {bn}, {epi}, {32 if wide else 16}, {bfm}, {walk}>"

    if agg:
        # the roofline kernel is the matrix-core kernel with the largest share of step time.  `achieved` follows the
        # contract: ALGORITHMIC flops (SURVEY 8d counts a 3x3 conv as 9 taps) / kernel time.  A Winograd kernel executes
        # 2.25x fewer matrix FLOPs than that, so its algorithmic rate can exceed the fp32 MFMA peak; `executed_tflops`
        # (what the matrix pipe actually does) is reported next to it, and so is the largest direct instantiation.
        dom = max(agg, key=lambda v: agg[v][2])
        cnt, fl, ms = agg[dom]
        tot_ms = sum(v[2] for v in agg.values())
        tot_fl = sum(v[1] for v in agg.values())
        exec_scale = (1.0 / 2.25) if dom >= 20000 else 1.0
        roof = dict(bound="mfma", kernel=vname(dom),
                    achieved=fl / (ms * 1e-3) / 1e12, peak=peak_tflops, unit="TFLOP/s",
                    frac=fl / (ms * 1e-3) / 1e12 / peak_tflops, traffic=pmc_traffic(dom) if args.dtype == "f32" else None,
                    executed_tflops=fl * exec_scale / (ms * 1e-3) / 1e12,
                    launches_per_step=cnt, avg_launch_us=ms * 1e3 / cnt, flops_per_launch=fl / cnt)
        if dom >= 20000:
            roof["note"] = ("Winograd F(2x2,3x3): algorithmic (direct-conv) flops / time; the kernel executes 2.25x fewer "
                            "MFMA flops (executed_tflops), see DESIGN.md 3.1b")
            d2 = max((v for v in agg if v < 20000), key=lambda v: agg[v][2])
            c2, f2, m2 = agg[d2]
            roof["largest_direct_kernel"] = dict(kernel=vname(d2), achieved=f2 / (m2 * 1e-3) / 1e12,
                                                 frac=f2 / (m2 * 1e-3) / 1e12 / peak_tflops, launches_per_step=c2,
                                                 avg_launch_us=m2 * 1e3 / c2, traffic=pmc_traffic(d2) if args.dtype == "f32" else None)
        roof["igemm_family"] = dict(achieved=tot_fl / (tot_ms * 1e-3) / 1e12, ms_per_step=tot_ms, flops_per_step=tot_fl,
                                    by_variant={vname(v, short=True): dict(launches=a[0], ms=round(a[2], 3),
                                                                           tflops=round(a[1] / (a[2] * 1e-3) / 1e12, 2))
                                                for v, a in sorted(agg.items())})

    # ---- the split-bf16 matrix-core modes next to the exact one (same model / batch / timing protocol; reported beside
    # `value`, never as it) ----
    alt = None
    if args.dtype == "f32" and not args.no_alt_modes:
        what = {"f32x3": "split-bf16 products (hi*hi + hi*lo + lo*hi, ~2^-16), fp32 accumulate and storage; "
                         "ds6g_set_compute_mode(2); Winograd off (direct implicit GEMM); tests/test_bf16_gpu.py",
                "f32x6": "three-way bf16 split, six products (fp32-grade, ~2^-23) in the direct kernels and attention, fp32 "
                         "Winograd kept for the 3x3/1 convs; ds6g_set_compute_mode(3); tests/test_bf16_gpu.py"}
        alt = {}
        for mode in ("f32x3", "f32x6"):
            try:
                with torch.no_grad():
                    model.eval()
                    ops.set_compute_mode("f32")
                    ref_logits = model(fronts, lidars, radars, gps).float().clone()
                    ops.set_compute_mode(mode)
                    m_logits = model(fronts, lidars, radars, gps).float()
                    dev_rel = float((m_logits - ref_logits).abs().max() / ref_logits.abs().max())
                    model.train()
                for _ in range(2):
                    train_iteration(model, opt, batch, ema, reducer)
                sync()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    train_iteration(model, opt, batch, ema, reducer)
                sync()
                el = time.perf_counter() - t0
                if world > 1:
                    t = torch.tensor([el], device=dev, dtype=torch.float64)
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    el = float(t.item())
                alt[mode] = {"value": args.batch * world * args.steps / el, "unit": "samples/s",
                             "ms_per_step": el / args.steps * 1e3, "eval_logits_max_dev_vs_f32_rel": dev_rel,
                             "what": what[mode]}
            except Exception as e:  # an extra mode never takes the headline line down with it
                alt[mode] = {"error": repr(e)}
            finally:
                ops.set_compute_mode("f32")
                model.train()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_batch, args.cpu_steps)

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
                       "parallelism": f"dp{world}", "ema": bool(args.ema)},
            "loss": final_loss,
            "algorithmic_gflop_per_sample": 559.3,
            "model_tflops": value * 559.3e9 / 1e12,
            "roofline": roof,
            "cpu_baseline": cpu,
            "other_modes": alt,
        }
        if cpu:
            out["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
