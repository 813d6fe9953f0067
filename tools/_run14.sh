cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04n; mkdir -p $O
DTYPE=f32 python tools/host_profile.py > $O/hostprof_f32.txt 2>&1
head -60 $O/hostprof_f32.txt | cut -c1-150
