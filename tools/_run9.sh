cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bgemm_gpu.py tests/test_ops_gpu.py tests/test_bench_shapes_gpu.py -x -q -k "not timed_configuration and not full_path and not config" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
DS6G_LIB=deepsense6g_tii_amd/libds6g_gemmclk.so WHICH=f32,bf16 timeout -k 10 300 python tools/gemm_clocks.py > $O/gemm_clocks.txt 2>&1
grep -A2 "N=2048 K=512\|N=512 K=512" $O/gemm_clocks.txt | grep -v "^--" | cut -c1-330
for d in f32 bf16; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba --dtype $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['ms_per_step'])" >> $O/bench.txt; done
cat $O/bench.txt
