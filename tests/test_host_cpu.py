"""CPU: host-side logic of the product (no kernel is launched): parameter naming / arena order, schedule,
metrics, synthetic data contract, loud failure without a GPU, and the data-parallel gradient reducer on a
world_size-2 gloo group."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from deepsense6g_tii_amd import dist as ddist
from deepsense6g_tii_amd import train as T
from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
from deepsense6g_tii_amd.synthetic import make_batch, soft_beam_target
from oracle import fusion_ref as fr
from oracle import train_ref as tr


def test_state_dict_names_and_shapes_equal_reference_contract():
    m = TransFuser(GlobalConfig(), "cpu")
    ref = fr.param_shapes(fr.RefConfig())
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    # a reference-format checkpoint (incl. DataParallel 'module.' prefix stripped) loads strictly
    m.load_state_dict(fr.make_state(fr.RefConfig(), seed=1), strict=True)


def test_config_mirrors_reference_defaults():
    c = GlobalConfig(n_layer=2)
    assert (c.seq_len, c.n_views, c.vert_anchors, c.horz_anchors, c.n_head, c.block_exp) == (5, 1, 8, 8, 4, 4)
    assert c.n_layer == 2 and c.embd_pdrop == c.attn_pdrop == c.resid_pdrop == 0.1


def test_no_cpu_fallback():
    m = TransFuser(GlobalConfig(n_layer=1), "cpu")
    f, l, r, g, t, _ = make_batch(1, res=256)
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        m(f, l, r, g)


def test_arena_milestones_are_contiguous_and_in_backward_order():
    names = [n for n, _ in TransFuser(GlobalConfig(), "cpu").named_parameters()]
    ms = sorted(names, key=TransFuser._milestone)
    order = [TransFuser._milestone(n) for n in ms]
    assert order == sorted(order) and set(order) == set(range(10))
    assert TransFuser._milestone("join.0.weight") == 0
    assert TransFuser._milestone("encoder.transformer4.blocks.0.ln1.weight") == 1
    assert TransFuser._milestone("encoder.vel_emb4.bias") == 1
    assert TransFuser._milestone("encoder.image_encoder.features.layer4.2.conv2.weight") == 2
    assert TransFuser._milestone("encoder.transformer1.pos_emb") == 7
    assert TransFuser._milestone("encoder.radar_encoder._model.layer1.0.bn1.bias") == 8
    assert TransFuser._milestone("encoder.lidar_encoder._model.conv1.weight") == 9
    assert TransFuser._milestone("encoder.image_encoder.features.bn1.weight") == 9


def test_schedule_and_metrics_equal_oracle():
    class Opt:
        param_groups = [dict(lr=1e-4)]
    sch = T.CyclicCosineDecayLR(Opt())
    for e in range(0, 60):
        assert abs(sch.get_last_lr()[0] - tr.cyclic_cosine_lr(e, 1e-4)) < 1e-15, e
        sch.step()
    rng = np.random.default_rng(0)
    scores = rng.standard_normal((50, 64))
    pred = np.argsort(-scores, axis=1)
    y = rng.integers(0, 64, 50)
    assert abs(T.compute_DBA_score(pred, y) - tr.compute_dba_score(pred, y)) < 1e-12
    assert list(T.compute_acc(pred, y)) == list(tr.compute_acc(pred, y))


def test_synthetic_batch_contract():
    f, l, r, g, t, beam = make_batch(3, seed=5)
    assert len(f) == 5 and len(l) == 5 and len(r) == 5
    assert f[0].shape == (3, 3, 256, 256) and l[0].shape == (3, 1, 256, 256) and r[0].shape == (3, 2, 256, 256)
    assert g.shape == (3, 2, 2) and t.shape == (3, 64)
    assert f[0].min() >= 0 and f[0].max() <= 255 and (f[0] == f[0].round()).all()
    assert {round(float(v), 1) for v in np.unique(l[0].numpy())} <= {0.0, 0.2, 0.4, 0.6, 0.8, 1.0}
    assert (g[:, :, 0] == g[:, :, 1]).all()
    assert (soft_beam_target(beam) - fr.soft_beam_target(beam)).abs().max() == 0
    # target peak 1.25 * pdf(0; sigma .5) at the beam index, support <= 11 beams (data2_seq.py:162-167)
    assert abs(float(t[0, beam[0]]) - 1.25 / (0.5 * np.sqrt(2 * np.pi))) < 1e-5
    assert int((t[0] > 0).sum()) <= 11


def test_grad_reducer_bucketing_single_process():
    g = torch.arange(100, dtype=torch.float32)
    red = ddist.GradReducer(g, min_bucket_elems=30)
    red.begin()
    red.ready(0, 0, 10)
    red.ready(1, 10, 25)
    assert red.issued == []            # below the bucket threshold: coalesce
    red.ready(2, 25, 60)
    assert red.issued == [(0, 60)]
    red.ready(3, 60, 70)
    red.finish()
    assert red.issued == [(0, 60), (60, 70)]
    with pytest.raises(AssertionError):
        red.begin()
        red.ready(0, 5, 10)             # not a contiguous prefix


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = ddist.init_distributed("gloo")
    assert (r, w) == (rank, world)
    n = 1000
    grads = torch.full((n,), float(rank + 1))
    grads[:10] += torch.arange(10.0) * (rank + 1)
    red = ddist.GradReducer(grads, min_bucket_elems=300)
    red.begin()
    for k, (lo, hi) in enumerate([(0, 100), (100, 350), (350, 900), (900, 1000)]):
        red.ready(k, lo, hi)
    red.finish()
    # parameters: rank 1 starts different, broadcast makes them equal
    class M:
        def __init__(self):
            self.p = torch.full((16,), float(rank))
            self.b = torch.full((4,), float(rank))
        def flat_parameters(self):
            return self.p, None
        def buffers(self):
            return [self.b]
    m = M()
    ddist.broadcast_parameters(m)
    out[rank] = (grads.clone(), red.issued, m.p.clone(), m.b.clone())
    torch.distributed.destroy_process_group()


def test_data_parallel_allreduce_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    g0, issued0, p0, b0 = out[0]
    g1, issued1, p1, b1 = out[1]
    expect = torch.full((1000,), 3.0)
    expect[:10] += torch.arange(10.0) * 3
    assert torch.equal(g0, expect) and torch.equal(g1, expect)       # SUM over ranks, identical on both
    assert issued0 == issued1 == [(0, 350), (350, 900), (900, 1000)]  # coalesced to >= 300-element buckets
    assert torch.equal(p0, torch.zeros(16)) and torch.equal(p1, torch.zeros(16))
    assert torch.equal(b1, torch.zeros(4))


# the bucket table of the REAL model: 78 422 528 gradients, arena in backward-completion order, milestones
# coalesced to >= 8 M elements, everything up to the first GPT stage flushed at milestone 7
REAL_BUCKETS = [26016704, 29901824, 17619712, 4347200, 537088]


def _real_milestones():
    m = TransFuser(GlobalConfig(), "cpu")
    _, pslice, ends, total = m.arena_layout()
    assert total == sum(p.numel() for p in m.parameters()) == 78422528      # no padding holes in the real model
    return [ends[k] for k in range(10)], total


def test_real_model_bucket_table_single_process():
    ends, total = _real_milestones()
    red = ddist.GradReducer(torch.zeros(1), min_bucket_elems=8 << 20, flush_at=7)    # attach()'s arguments
    red.g = torch.empty(0)          # slices of an empty tensor: only the bookkeeping is exercised here
    red.begin()
    lo = 0
    for k, hi in enumerate(ends):
        red.ready(k, lo, hi)
        lo = hi
    red.finish()
    assert [hi - lo for lo, hi in red.issued] == REAL_BUCKETS and red.issued[-1][1] == total


def _dp_worker_real(rank, world, port, ends, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, _ = ddist.init_distributed("gloo")
    assert (r, w) == (rank, world)
    # small integers: the SUM over ranks is exact in fp32 whatever order gloo adds in
    base = (torch.arange(total, dtype=torch.int32) % 251).to(torch.float32)
    grads = base + float(rank)
    red = ddist.GradReducer(grads, min_bucket_elems=8 << 20, flush_at=7)
    red.begin()
    lo = 0
    for k, hi in enumerate(ends):       # the backward walk's milestone order
        red.ready(k, lo, hi)
        lo = hi
    red.finish()
    expect = base * world + float(world * (world - 1) // 2)
    out[rank] = (bool(torch.equal(grads, expect)), list(red.issued), float(grads.double().sum()))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_real_model_bucket_table_allreduce_gloo(world):
    """the gradient exchange of `bench.py --gpus 4 / 8` - the real arena size and the real bucket
    table, world_size 4 and 8 over gloo on the CPU: every rank issues the same five buckets and ends with the exact sum."""
    ends, total = _real_milestones()
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_dp_worker_real, args=(r, world, port, ends, total, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    sums = set()
    for r in range(world):
        ok, issued, total_sum = out[r]
        assert ok, r
        assert [hi - lo for lo, hi in issued] == REAL_BUCKETS
        sums.add(total_sum)
    assert len(sums) == 1


def test_checkpoint_prefix_and_csv(tmp_path):
    sd = {"module.join.0.weight": 1, "encoder.vel_emb1.bias": 2}
    assert T.strip_module_prefix(sd) == {"join.0.weight": 1, "encoder.vel_emb1.bias": 2}
    pred = np.array([[4, 3, 50] + [0] * 61, [63, 0, 1] + [0] * 61])
    out = tmp_path / "beam_pred.csv"
    T.save_pred_to_csv(pred, target_csv=str(out))
    lines = out.read_text().strip().split("\n")
    assert lines[0] == "index,top-1 beam,top-2 beam,top-3 beam"   # train2_seq.py:343-346 (1-based beams)
    assert lines[1] == "0,5,4,51" and lines[2] == "1,64,1,2"


def test_grad_reducer_flushes_early_at_the_given_milestone():
    """flush_at: the bucket open at that milestone goes out even below the size threshold, so the last (unoverlappable)
    bucket only holds what comes after it."""
    import torch
    from deepsense6g_tii_amd import dist as ddist
    g = torch.zeros(100)
    red = ddist.GradReducer(g, min_bucket_elems=1000, flush_at=2)
    red.begin()
    red.ready(0, 0, 10)
    red.ready(1, 10, 30)
    assert red.issued == []
    red.ready(2, 30, 60)
    assert red.issued == [(0, 60)]
    red.ready(3, 60, 90)
    red.finish()
    assert red.issued == [(0, 60), (60, 90)]


def test_dropout_seed_differs_per_rank_and_rng_state_roundtrips():
    """data-parallel ranks must not share dropout masks; the (base seed, counter) state is checkpointed.
    ADVICE r02: resuming a data-parallel run from rank 0's checkpoint must NOT put every rank on rank 0's mask stream -
    the checkpoint carries the unmixed base seed, each rank re-applies its own mix, in either call order."""
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    m = TransFuser(GlobalConfig(n_layer=1), "cpu")
    base = m._seed
    seeds = [m.set_dropout_seed(base, r) for r in range(8)]
    assert seeds[0] == base and len(set(seeds)) == 8 and all(0 <= s < 2 ** 64 for s in seeds)
    # rank 0 checkpoints after 12345 training forwards
    m0 = TransFuser(GlobalConfig(n_layer=1), "cpu")
    m0.set_dropout_seed(base, 0)
    m0._salt_host = 12345 * m0.SALT_STRIDE
    st = m0.rng_state()
    assert st == dict(seed=base, counter=12345 * m0.SALT_STRIDE)
    # rank 3 resumes from it: attach() (= set_dropout_seed(base_seed, rank)) before the load ...
    m3 = TransFuser(GlobalConfig(n_layer=1), "cpu")
    m3.set_dropout_seed(m3._base_seed, 3)
    m3.set_rng_state(st)
    assert (m3._seed, m3._salt_host) == (seeds[3], 12345 * m0.SALT_STRIDE)
    # ... or after it (dist.attach mixes from the UNMIXED base seed, so mixing happens exactly once)
    m3b = TransFuser(GlobalConfig(n_layer=1), "cpu")
    m3b.set_rng_state(st)
    assert m3b._seed == base
    m3b.set_dropout_seed(m3b._base_seed, 3)
    m3b.set_dropout_seed(m3b._base_seed, 3)          # attach() twice is harmless
    assert (m3b._seed, m3b._salt_host) == (seeds[3], 12345 * m0.SALT_STRIDE)
    # a rank-3 checkpoint holds the base seed too, so any rank may resume from any rank's file
    assert m3.rng_state() == st


def test_per_scenario_metrics_and_confidence_csv(tmp_path):
    """train2_seq.py:195-207 (per-scenario accuracy / DBA) and :243-252 (softmax max-confidence file)."""
    rng = np.random.default_rng(0)
    pred = np.stack([rng.permutation(64) for _ in range(12)])
    truth = rng.integers(0, 64, 12)
    scen = np.array(["scenario31"] * 5 + ["scenario33"] * 7)
    table = T.per_scenario_metrics(pred, truth, scen)
    assert set(table) == {"scenario31", "scenario33"}                 # scenarios without samples are skipped (:199)
    acc31, dba31 = table["scenario31"]
    assert list(acc31) == list(tr.compute_acc(pred[:5], truth[:5])) and abs(dba31 - tr.compute_dba_score(pred[:5], truth[:5])) < 1e-12
    acc33, dba33 = table["scenario33"]
    assert list(acc33) == list(tr.compute_acc(pred[5:], truth[5:])) and abs(dba33 - tr.compute_dba_score(pred[5:], truth[5:])) < 1e-12

    class Dummy(torch.nn.Module):          # stands in for the HIP model: test() only needs a callable with train()/eval()
        def forward(self, a, b, c, d):
            return a
    logits = torch.randn(5, 64, generator=torch.Generator().manual_seed(1))
    p, c = T.test(Dummy(), [(logits[:2], None, None, None), (logits[2:], None, None, None)],
                  target_csv=str(tmp_path / "beam_pred.csv"), confidence_csv=str(tmp_path / "conf.csv"))
    assert np.array_equal(p, torch.argsort(logits, dim=1, descending=True).numpy())
    assert np.allclose(c, torch.softmax(logits, 1).max(1)[0].numpy())
    rows = (tmp_path / "conf.csv").read_text().strip().split("\n")
    assert rows[0] == ",0" and len(rows) == 6 and abs(float(rows[3].split(",")[1]) - float(c[2])) < 1e-7
    assert (tmp_path / "beam_pred.csv").read_text().startswith("index,top-1 beam")


def test_bench_roofline_fraction_is_an_executed_fraction():
    """roofline.frac must be executed matrix FLOPs / time / peak (<= 1), the algorithmic rate a separate key."""
    import bench
    # a Winograd record: 110 launches, algorithmic 18.12 GFLOP each, 110.2 us each (profiles/r01 numbers)
    agg = {20000: [110, 110 * 18119393280.0, 110 * 0.1102], 12: [130, 130 * 6.6e9, 130 * 0.0663]}
    roof = bench.roofline_from_records(agg, "f32", traffic=lambda v: None)
    assert roof["kernel"] == "winograd_fwd_kernel<1>"
    assert 0.4 < roof["frac"] < 0.5 and abs(roof["achieved"] - roof["frac"] * roof["peak"]) < 1e-9
    assert roof["algorithmic_tflops"] > roof["achieved"] * 2.2
    assert roof["frac"] <= 1.0 and roof["largest_direct_kernel"]["frac"] <= 1.0


def test_train_pieces_match_reference_run(tmp_path):
    """The product's host-side training pieces (train.py) against outputs of the REFERENCE's own functions
    (tests/golden/train_golden.npz <- tests/golden/make_golden_train.py: train2_seq.py:338-383, scheduler.py:82-119)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_golden.npz"))
    for base in (1e-4, 5e-4):
        class Opt:
            param_groups = [dict(lr=base)]
        sch = T.CyclicCosineDecayLR(Opt())       # stepped once per epoch, as train2_seq.py:613-615
        for e in range(61):
            assert sch.get_last_lr()[0] == g[f"lr_base{base:g}"][e], (base, e)
            sch.step()
    for i in range(4):
        pred, true = g[f"metric{i}_pred"], g[f"metric{i}_true"]
        assert np.array_equal(T.compute_acc(pred, true), g[f"metric{i}_acc"])
        assert np.array_equal(T.compute_acc(pred, true, top_k=(1, 3, 5)), g[f"metric{i}_acc5"])
        assert abs(T.compute_DBA_score(pred, true) - float(g[f"metric{i}_dba"])) < 1e-12
        assert abs(T.compute_DBA_score(pred, true, max_k=5, delta=3) - float(g[f"metric{i}_dba_k5_d3"])) < 1e-12
    out = tmp_path / "beam_pred.csv"
    T.save_pred_to_csv(g["metric1_pred"], target_csv=str(out))
    assert out.read_text() == str(g["csv_text"])            # byte-identical to the file pandas wrote for the reference
    T.save_pred_to_csv(g["metric1_pred"], top_k=(1, 2, 3, 4, 5), target_csv=str(out))
    assert out.read_text() == str(g["csv_text_top5"])
