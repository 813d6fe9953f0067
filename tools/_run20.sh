cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04t; mkdir -p $O
for m in 0 1 0 1; do DS6G_BF16_STEMS=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba --no-extra-legs --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BF16_STEMS=$m', d['value'], d['ms_per_step'], d['loss'])"; done | tee $O/ab.txt
timeout -k 10 1100 python -m pytest tests/test_bf16_gpu.py tests/test_train_gpu.py tests/test_bench_shapes_gpu.py -x -q -k "bf16 or config" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
