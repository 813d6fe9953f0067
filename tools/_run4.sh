cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_bgemm_gpu.py -x -q -k "fused_batchnorm or conv_fwd" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
for m in 0 1; do DS6G_FUSE_BN_STATS16=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FUSE_BN_STATS16=$m', d['value'], d['ms_per_step'])" >> $O/bench_bf16.txt; done
cat $O/bench_bf16.txt
timeout -k 10 500 python -m pytest tests/test_bf16_gpu.py tests/test_train_gpu.py -x -q > $O/pytest2.txt 2>&1 || { tail -30 $O/pytest2.txt; exit 1; }
tail -3 $O/pytest2.txt
