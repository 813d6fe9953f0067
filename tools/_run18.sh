cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04r; mkdir -p $O
for m in 0 1 0 1; do DS6G_WINO_BATCHW=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BATCHW=$m', d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
timeout -k 10 1100 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
