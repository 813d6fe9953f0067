cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04e; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gru_head.py tests/test_bench_shapes_gpu.py -x -q -s -k "30to5 or 1922 or gru" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -25 $O/pytest.txt
