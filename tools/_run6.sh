cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04f; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_input_gpu.py -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -5 $O/pytest.txt
