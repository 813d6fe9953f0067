cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04q; mkdir -p $O
for d in f32 bf16; do DTYPE=$d python tools/host_overhead.py 2>/dev/null | tail -2; done | tee $O/host.txt
timeout -k 10 1100 python -m pytest tests/test_train_gpu.py tests/test_model_gpu.py tests/test_gru_head.py -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
