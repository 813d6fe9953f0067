cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p; mkdir -p $O
for d in f32 bf16; do DTYPE=$d python tools/host_overhead.py 2>/dev/null | tail -2; done | tee $O/host.txt
