"""Feasibility probe: the GPT block chain (8 blocks fwd + bwd) of one stage at bs=12 on ONE stream vs as two half-batch
chains on TWO streams (kernels of the two halves overlap: attention (VALU / softmax heavy) against linears (MFMA heavy),
and each other's launch ramps / tails).  Prints ms per chain for C in 64..512."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd.model import GlobalConfig, TransFuser

dev = torch.device("cuda:0")
cfg = GlobalConfig()
model = TransFuser(cfg, dev)
model.train()
model._recording = True
B, T = 12, 962
streams = [torch.cuda.Stream(dev) for _ in range(2)]
from deepsense6g_tii_amd import ops
for st in streams:
    model._ws_side[st.cuda_stream] = ops.Workspace(dev, 1 << 30)


def chain(gpt, x, Bh, wg=True):
    ctxs = []
    for blk in gpt.blocks:
        x, c = model._gpt_block_fwd(blk, x, Bh, T, True)
        ctxs.append(c)
    dx = torch.ones_like(x)
    for blk, c in zip(reversed(list(gpt.blocks)), reversed(ctxs)):
        dx, _ = model._gpt_block_bwd(blk, c, dx, Bh, T)
    return dx


for s in (1, 2, 3, 4):
    gpt = getattr(model.encoder, f"transformer{s}")
    C = gpt.n_embd
    x = torch.randn(B * T, C, device=dev)
    for mode in ("one", "two"):
        for p in model.parameters():
            p.grad = None
        model._begin_backward()
        model.overlap_wgrad = True

        def run():
            if mode == "one":
                chain(gpt, x, B)
                model._wg_join()
            else:
                cur = torch.cuda.current_stream()
                for h, st in enumerate(streams):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        chain(gpt, x[h * (B // 2) * T:(h + 1) * (B // 2) * T], B // 2)
                        model._wg_join()
                for st in streams:
                    cur.wait_stream(st)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        print(f"stage {s} C={C} {mode}-stream: {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)
