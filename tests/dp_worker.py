"""Child process of tests/test_dp_gpu.py: ONE data-parallel rank running one train_iteration of a small TransFuser
through dist.attach (bucketed all-reduce issued from the backward walk), then dumping its gradient arena, parameters and
bucket list.  Launched fresh (nothing touches the GPU before init_distributed); on a one-GPU box the ranks share the card
through the rehearsal knobs DS6G_DIST_BACKEND=gloo / DS6G_FORCE_DEVICE=0 (RCCL refuses two ranks on one device)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shard(rank, world, rcfg, seed):
    """rank's slice of a world-sized batch, split on dim 0 exactly as DataParallel.scatter does (train2_seq.py:538)"""
    from oracle import fusion_ref as fr
    imgs, lids, rads, gps, target, _ = fr.make_inputs(rcfg, world, seed=seed)
    cut = lambda seq: [t[rank:rank + 1].contiguous() for t in seq]  # noqa: E731
    return cut(imgs), cut(lids), cut(rads), gps[rank:rank + 1].contiguous(), target[rank:rank + 1].contiguous()


def main():
    out_dir, seed, dropout = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    import torch
    from deepsense6g_tii_amd import dist as ddist
    rank, world, local = ddist.init_distributed()      # first GPU-touching call of this process
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
    from deepsense6g_tii_amd.train import FusedAdamW, train_iteration
    from oracle import fusion_ref as fr
    kw = dict(n_layer=1, embd_pdrop=dropout, attn_pdrop=dropout, resid_pdrop=dropout)
    rcfg = fr.RefConfig(**kw)
    model = TransFuser(GlobalConfig(**kw), dev)
    if rank == 0:   # rank 0 holds the real weights, the others get them by the start-up broadcast
        model.load_state_dict(fr.make_state(rcfg, seed=seed))
    model.train()
    opt = FusedAdamW(model, lr=1e-3)
    ddist.broadcast_parameters(model)
    seed0 = model._seed
    red = ddist.attach(model, opt, min_bucket_elems=2 << 20)
    batch = shard(rank, world, rcfg, seed + 1)
    loss, logits = train_iteration(model, opt, batch, None, red)
    torch.cuda.synchronize()
    p, g = model.flat_parameters()
    torch.save(dict(grad=g.cpu(), param=p.cpu(), issued=list(red.issued), loss=float(loss), world=world,
                    grad_scale=opt.grad_scale, seed0=int(seed0), seed=int(model._seed), logits=logits.cpu(),
                    milestone_end=dict(model._milestone_end), used=int(model._arena_used)),
               os.path.join(out_dir, f"rank{rank}.pt"))
    # misuse guard: a second backward without zero_grad would re-reduce an already reduced arena -> must raise
    try:
        model.train_step_loss(*batch)
        guarded = False
    except RuntimeError:
        guarded = True
    open(os.path.join(out_dir, f"guard{rank}.txt"), "w").write(str(guarded))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
