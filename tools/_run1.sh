cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04a
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba > gpurun_out/r04a/bench_f32.json 2> gpurun_out/r04a/bench_f32.err &&
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --no-dba --dtype bf16 > gpurun_out/r04a/bench_bf16.json 2> gpurun_out/r04a/bench_bf16.err &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04a/tr_bf16 -o t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt-modes --no-dba --dtype bf16 > gpurun_out/r04a/tr_bf16.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04a/tr_f32 -o t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt-modes --no-dba > gpurun_out/r04a/tr_f32.log 2>&1
cat gpurun_out/r04a/bench_f32.json gpurun_out/r04a/bench_bf16.json | cut -c1-400
