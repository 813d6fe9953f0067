"""Data parallelism end to end on ONE card (replaces torch.nn.DataParallel, train2_seq.py:538):
two fresh child processes = two ranks, each running one train_iteration of a small TransFuser with dist.attach
(TransFuser._milestone_done -> GradReducer.ready -> bucketed all-reduce during the backward walk).  Checked:
  * both ranks' gradient arenas equal the SUM of the two single-process shard arenas (per-shard BatchNorm);
  * the buckets issued cover the arena exactly once, in order, ending on milestone boundaries;
  * parameters after the AdamW step with grad_scale = 1/2 are identical on both ranks and equal the oracle AdamW on the
    averaged gradient;
  * ranks draw different dropout seeds; a backward without zero_grad raises instead of double-reducing."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(tmp_path, seed, dropout, world=2):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DS6G_DIST_BACKEND="gloo", DS6G_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path), str(seed),
                                       str(dropout)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for rank, p in enumerate(procs):
        assert p.returncode == 0, f"rank {rank} failed:\n{outs[rank][-4000:]}"
    return [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True) for r in range(world)]


def test_two_ranks_reduce_to_the_sum_of_shard_gradients(dev, tmp_path):
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from oracle import fusion_ref as fr
    from oracle import train_ref as tr
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dp_worker import shard
    seed = 71
    ranks = _launch(tmp_path, seed, 0.0)
    # single-process shard runs (the reference semantics of one DataParallel replica each)
    kw = dict(n_layer=1, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    rcfg = fr.RefConfig(**kw)
    sd = fr.make_state(rcfg, seed=seed)
    single = []
    for r in range(2):
        model = TransFuser(GlobalConfig(**kw), dev)
        model.load_state_dict(sd)
        model.train()
        loss, _ = model.train_step_loss(*shard(r, 2, rcfg, seed + 1))
        torch.cuda.synchronize()
        p, g = model.flat_parameters()
        single.append(dict(grad=g.cpu().clone(), loss=float(loss)))
        param0 = p.cpu().clone()      # train_step_loss does not touch the parameters: still the start-up weights
        del model
    gsum = single[0]["grad"] + single[1]["grad"]
    scale = gsum.abs().max().item()
    for r in range(2):
        assert ranks[r]["world"] == 2 and ranks[r]["grad_scale"] == 0.5
        assert abs(ranks[r]["loss"] - single[r]["loss"]) < 1e-6 * abs(single[r]["loss"]) + 1e-9
        err = (ranks[r]["grad"] - gsum).abs().max().item()
        assert err <= 1e-5 * scale, (r, err, scale)
    assert torch.equal(ranks[0]["grad"], ranks[1]["grad"])          # bitwise: same sum on both ranks
    assert torch.equal(ranks[0]["param"], ranks[1]["param"])        # replicas stay in lock-step
    # AdamW on the averaged gradient (grad_scale = 1/world folded into the kernel)
    ref = param0.clone()
    m, v = torch.zeros_like(ref), torch.zeros_like(ref)
    tr.adamw_step(ref, ranks[0]["grad"] * 0.5, m, v, 1, 1e-3)
    assert (ranks[0]["param"] - ref).abs().max().item() < 1e-7 + 2e-6 * ref.abs().max().item()
    # bucket list: contiguous, in order, covers [0, arena_used) exactly once, every cut on a milestone boundary
    issued, used = ranks[0]["issued"], ranks[0]["used"]
    assert issued == ranks[1]["issued"] and len(issued) >= 3
    assert issued[0][0] == 0 and issued[-1][1] == used
    for (lo, hi), (lo2, hi2) in zip(issued, issued[1:]):
        assert hi == lo2 and hi > lo
    ends = set(ranks[0]["milestone_end"].values())
    assert all(hi in ends for _, hi in issued)
    # guard against double reduction
    for r in range(2):
        assert open(os.path.join(tmp_path, f"guard{r}.txt")).read() == "True"


def test_ranks_draw_independent_dropout_masks(dev, tmp_path):
    ranks = _launch(tmp_path, 72, 0.1)
    assert ranks[0]["seed0"] == ranks[1]["seed0"]
    assert ranks[0]["seed"] == ranks[0]["seed0"]                    # rank 0 keeps the single-process stream
    assert ranks[1]["seed"] != ranks[0]["seed"]
    assert torch.isfinite(ranks[0]["grad"]).all() and torch.equal(ranks[0]["grad"], ranks[1]["grad"])


def test_rccl_process_group_orders_buckets_after_their_producers(dev, tmp_path):
    """the gradient exchange on the REAL RCCL backend (world size 1: one GPU here), multi_stream and
    overlap_wgrad on - see tests/nccl_worker.py for what is checked and why only ProcessGroupNCCL exercises it."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               DS6G_DIST_FORCE_INIT="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DS6G_DIST_BACKEND", None)
    env.pop("DS6G_FORCE_DEVICE", None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "nccl_worker.py"), str(tmp_path)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        out, _ = p.communicate(timeout=600)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    assert p.returncode == 0, out.decode(errors="replace")[-4000:]
    r = torch.load(os.path.join(tmp_path, "nccl_rank0.pt"), weights_only=True)
    assert r["backend"] == "nccl" and r["grad_scale"] == 1.0
    assert r["nan_plain"] == 0 and r["nan_reduced"] == 0          # every gradient of the NaN-poisoned arena was written
    assert r["same"]                                                # all-reduce at world 1 = identity, bit for bit
    assert r["loss_a"] == r["loss_b"] == r["loss_c"]
    issued, used = r["issued"], r["used"]
    assert len(issued) >= 3 and r["n_probes"] == len(issued)
    assert issued[0][0] == 0 and issued[-1][1] == used
    for (lo, hi), (lo2, hi2) in zip(issued, issued[1:]):
        assert hi == lo2 and hi > lo
    assert all(hi in set(r["milestone_end"].values()) for _, hi in issued)
    # every bucket was FINAL when it was handed to the process group: the probe taken under RCCL's own ordering (after the
    # calling stream at issue time) holds the final values - no producer on a side stream was still running or unjoined
    assert r["nan_probe"] == [0] * len(issued), r["nan_probe"]
    assert all(r["ok_probe"]), r["ok_probe"]
    assert r["params_finite"]


def test_persistent_winograd_kernel_next_to_resident_channel_workgroups(dev):
    """Pre-validation of the 8-GPU run (no multi-GPU node): under data parallelism RCCL keeps a few channel workgroups resident
    while the backward pass runs, and a winograd_pc_kernel workgroup needs a whole CU (512 threads x 256 VGPRs, 128 KiB of LDS) -
    a full-size grid would run its last workgroups in a second round (2x that launch).  The default grid (the smallest one that
    needs no more rounds than all CUs: 240 workgroups at the model's batch) leaves 16 CUs free: with 16 stand-in workgroups
    (ds6g_debug_occupy_cus, 40 KiB of LDS each: no room for a persistent workgroup beside them) resident on a side stream, no
    launch of the persistent kernel may take 1.6x its undisturbed time, and the results stay bit-identical."""
    import ctypes
    from deepsense6g_tii_amd import ops
    from deepsense6g_tii_amd._lib import lib
    L = lib()
    N, H, W, C, K = 60, 32, 32, 128, 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, C, generator=g).to(dev)
    w = (torch.randn(K, 3, 3, C, generator=g) * 0.05).to(dev)
    u = ops.winograd_weights(w.data_ptr(), K, C, dev)
    from deepsense6g_tii_amd.dist import concurrent_side_stream
    side, _keep = concurrent_side_stream(dev)     # a stream that really overlaps the current one (probed)
    CAP = 256

    def timed(reps):
        L.profile_begin(CAP)
        ys = [ops.conv3x3_winograd(x, u, K) for _ in range(reps)]
        var = (ctypes.c_int * CAP)(); fl = (ctypes.c_double * CAP)(); tm = (ctypes.c_float * CAP)()
        k = L.profile_end(var, fl, tm, CAP)
        t = sorted(tm[i] * 1e3 for i in range(k) if var[i] == 20002)
        assert len(t) == reps, (len(t), [var[i] for i in range(k)])
        return ys[-1], t

    timed(3)
    y0, t0 = timed(20)
    L.debug_occupy_cus(16, 40 * 1024, 300_000, side.cuda_stream)      # resident for 0.3 s beside the launches below
    import time
    time.sleep(0.02)
    t_host = time.perf_counter()
    y1, t1 = timed(20)
    torch.cuda.current_stream().synchronize()
    assert time.perf_counter() - t_host < 0.15, "the launches waited for the occupier: not a co-residency measurement"
    # the same with a FULL-size grid's worth of resident workgroups missing: 32 occupied CUs leave 224 < 240 -> a second round
    L.debug_occupy_cus(16, 40 * 1024, 200_000, side.cuda_stream)
    y2, t2 = timed(10)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(y0, y2)
    print(f"with 32 resident workgroups: {t2[len(t2) // 2]:.1f} us median")
    med0, med1, worst1 = t0[len(t0) // 2], t1[len(t1) // 2], t1[-1]
    print(f"winograd_pc_kernel 60x32x32x128: {med0:.1f} us alone, {med1:.1f} us median / {worst1:.1f} us worst beside 16 resident workgroups")
    assert med1 < 1.6 * med0 and worst1 < 2.0 * med0, (med0, med1, worst1)
