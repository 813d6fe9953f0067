"""Child process of tests/test_dp_gpu.py::test_rccl_process_group_orders_buckets_after_their_producers: ONE rank on the
REAL RCCL backend (torch.distributed "nccl", world size 1 - RCCL refuses two ranks on one device, and this box has one).

What only RCCL exercises and the gloo rehearsal cannot: ProcessGroupNCCL runs a collective on its OWN stream, ordered after
the stream that is current when all_reduce() is called.  dist.GradReducer relies on exactly that: TransFuser joins the
trunk streams and the weight-gradient companion stream into the calling stream (_join / _wg_join) BEFORE _milestone_done
hands a bucket to the reducer.  Checked here, with multi_stream and overlap_wgrad on:
  * the gradient arena is poisoned with NaN before the step; a probe copies every bucket at issue time under the same
    ordering RCCL gets (an event on the current stream): a bucket sent before one of its producers had finished or been
    joined would carry NaN / stale values -> every probe must equal the final arena slice bit for bit;
  * the all-reduce really goes through ProcessGroupNCCL (force_collective; identity at world 1): the arena equals the run
    without a reducer bit for bit, and a following AdamW step consumes it on the calling stream (work.wait() ordering)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    import torch
    from deepsense6g_tii_amd import dist as ddist
    rank, world, local = ddist.init_distributed("nccl")      # first GPU-touching call of this process
    assert (rank, world) == (0, 1) and torch.distributed.is_initialized()
    assert torch.distributed.get_backend() == "nccl"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from deepsense6g_tii_amd.train import FusedAdamW, train_iteration
    from oracle import fusion_ref as fr
    kw = dict(n_layer=2)                     # reference dropout 0.1 / 0.1 / 0.1
    rcfg = fr.RefConfig(**kw)
    model = TransFuser(GlobalConfig(**kw), dev)
    model.load_state_dict(fr.make_state(rcfg, seed=91))
    model.train()
    assert model.multi_stream and model.overlap_wgrad
    opt = FusedAdamW(model, lr=1e-3)
    ddist.broadcast_parameters(model)        # no-op at world 1
    batch = fr.make_inputs(rcfg, 2, seed=92)[:5]
    p, g = model.flat_parameters()
    rng0 = model.rng_state()

    def one_backward():
        model.set_rng_state(rng0)            # same dropout masks in both runs
        opt.zero_grad(set_to_none=True)
        g.fill_(float("nan"))
        loss, _ = model.train_step_loss(*batch)
        torch.cuda.synchronize()
        return float(loss), g.clone()

    loss_a, plain = one_backward()
    red = ddist.attach(model, opt, min_bucket_elems=1 << 20)
    red.force_collective = True
    red.probe_stream = torch.cuda.Stream(dev)
    loss_b, reduced = one_backward()
    probes = [t.clone() for t in red.probes]
    issued = list(red.issued)
    n_works_used = len(issued)
    ok_probe = [bool(torch.equal(pr, reduced[lo:hi])) for pr, (lo, hi) in zip(probes, issued)]
    nan_probe = [int(torch.isnan(pr).sum()) for pr in probes]
    # a full iteration on top: AdamW reads the reduced arena on the calling stream right after finish()
    model.set_rng_state(rng0)
    loss_c, _ = train_iteration(model, opt, batch, None, red)
    torch.cuda.synchronize()
    torch.save(dict(loss_a=loss_a, loss_b=loss_b, loss_c=float(loss_c), same=bool(torch.equal(plain, reduced)),
                    nan_plain=int(torch.isnan(plain).sum()), nan_reduced=int(torch.isnan(reduced).sum()),
                    issued=issued, ok_probe=ok_probe, nan_probe=nan_probe, n_probes=len(probes),
                    used=int(model._arena_used), milestone_end=dict(model._milestone_end),
                    params_finite=bool(torch.isfinite(p).all()), backend=torch.distributed.get_backend(),
                    grad_scale=opt.grad_scale, n_collectives=n_works_used),
               os.path.join(out_dir, "nccl_rank0.pt"))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
