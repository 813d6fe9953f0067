cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04u; mkdir -p $O
DS6G_ATTN_FUSED128_BF16=1 timeout -k 10 600 python -m pytest tests/test_bgemm_gpu.py tests/test_bench_shapes_gpu.py -x -q -k "attention" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for m in 0 1 0 1; do DS6G_ATTN_FUSED128_BF16=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba --no-extra-legs --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FUSED128_BF16=$m', d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
