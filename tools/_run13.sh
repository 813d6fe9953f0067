cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04m; mkdir -p $O
for q in 8 16; do for r in 0 16 32; do GPU_MAX_HW_QUEUES=$q DS6G_PC_CU_RESERVE=$r timeout -k 10 200 python tools/coresidency.py 2>/dev/null | tail -1 | sed "s/^/queues=$q /" >> $O/coresidency.jsonl || exit 1; done; done
cat $O/coresidency.jsonl
GPU_MAX_HW_QUEUES=8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues=8', d['value'], d['ms_per_step'])"
