cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_bgemm_gpu.py tests/test_bench_shapes_gpu.py -x -q -k "attention" 2>&1 | tail -2
DS6G_ATTN_FUSED128_BF16=0 timeout -k 10 600 python -m pytest tests/test_bgemm_gpu.py -x -q -k "attention" 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba --no-extra-legs --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16', round(d['value'],1), round(d['ms_per_step'],2))"; done
