cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04o; mkdir -p $O
DTYPE=f32 python tools/host_overhead.py 2>/dev/null | tail -1 > $O/host.txt
DTYPE=bf16 python tools/host_overhead.py 2>/dev/null | tail -1 >> $O/host.txt
cat $O/host.txt
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_dp_gpu.py -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for d in f32 bf16; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba --no-extra-legs --dtype $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['ms_per_step'])"; done
