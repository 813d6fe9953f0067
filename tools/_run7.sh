cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04g; mkdir -p $O
python tools/igemm_calls.py > $O/igemm_calls_f32.txt 2>&1
head -60 $O/igemm_calls_f32.txt
