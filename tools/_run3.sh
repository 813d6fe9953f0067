cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c; mkdir -p $O
DTYPE=f32 python tools/host_overhead.py > $O/host_f32.txt 2>&1
DTYPE=bf16 python tools/host_overhead.py > $O/host_bf16.txt 2>&1
DTYPE=bf16 python tools/host_profile.py > $O/hostprof_bf16.txt 2>&1
for m in 0 1; do DS6G_BG_WXCD=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('WXCD=$m', d['value'], d['ms_per_step'])" >> $O/bench_bf16_wxcd.txt; done
cat $O/host_f32.txt $O/host_bf16.txt $O/bench_bf16_wxcd.txt
