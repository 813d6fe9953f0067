cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bgemm_gpu.py -x -q -k "stem" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
