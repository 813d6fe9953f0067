"""Pre-validation of the 8-GPU run on one GPU: the bs=12 training step while `OCC` workgroups of a stand-in kernel
(ds6g_debug_occupy_cus: 256 threads, 40 KiB of LDS each - a persistent winograd_pc_kernel workgroup cannot share their CU)
stay resident on a side stream, as RCCL's channel workgroups do during the backward pass.  Prints one JSON line: ms per step
and the winograd_pc_kernel launch durations (mean / max, from the library's per-launch HIP events in a single-stream step)
for OCC in 0 / 16 / 32 under the DS6G_PC_CU_RESERVE of the environment."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsense6g_tii_amd import ops
from deepsense6g_tii_amd._lib import lib
from deepsense6g_tii_amd.model import GlobalConfig, TransFuser
from deepsense6g_tii_amd.synthetic import make_batch
from deepsense6g_tii_amd.train import FusedAdamW, train_iteration

dev = torch.device("cuda:0")
ops.set_compute_mode(os.environ.get("DTYPE", "f32"))
model = TransFuser(GlobalConfig(), dev); model.train()
opt = FusedAdamW(model, lr=1e-4)
batch = make_batch(12, seed=100, device=dev)[:5]
for _ in range(3): train_iteration(model, opt, batch)
torch.cuda.synchronize()
from deepsense6g_tii_amd.dist import concurrent_side_stream
side, _keep = concurrent_side_stream(dev)
L = lib()
CAP = 1 << 14
out = dict(reserve=int(os.environ.get("DS6G_PC_CU_RESERVE", "0")), dtype=os.environ.get("DTYPE", "f32"), runs=[])
for occ in (0, 16, 32):
    n = 10
    if occ:
        L.debug_occupy_cus(occ, 40 * 1024, 1_000_000, side.cuda_stream)   # resident for 1 s: covers the 10 + 1 steps below
        time.sleep(0.01)
    t0 = time.perf_counter()
    for _ in range(n): train_iteration(model, opt, batch)
    torch.cuda.current_stream().synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    # per-launch durations of the persistent Winograd kernel under the same occupier (single stream: clean event brackets)
    model.multi_stream = False
    L.profile_begin(CAP)
    train_iteration(model, opt, batch)
    var = (ctypes.c_int * CAP)(); fl = (ctypes.c_double * CAP)(); tm = (ctypes.c_float * CAP)()
    k = L.profile_end(var, fl, tm, CAP)
    model.multi_stream = True
    pc = [tm[i] * 1e3 for i in range(k) if var[i] == 20002]
    torch.cuda.synchronize()   # the occupier has left
    out["runs"].append(dict(occupied_workgroups=occ, ms_per_step=round(ms, 2), winograd_pc_launches=len(pc),
                            winograd_pc_mean_us=round(sum(pc) / max(1, len(pc)), 1), winograd_pc_max_us=round(max(pc or [0]), 1)))
print(json.dumps(out))
