#!/bin/bash
# two-rank rehearsal of `bench.py --gpus 2` on ONE card (gloo all-reduce, both ranks on device 0): the launch line the driver
# uses for N > 1, with the rehearsal knobs of dist.py.  usage (GPU box): bash tools/rehearse_dp2.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
DS6G_DIST_BACKEND=gloo DS6G_FORCE_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-modes --no-dba
