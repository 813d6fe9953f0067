cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04j; mkdir -p $O
DS6G_BG_TILE=2 timeout -k 10 600 python -m pytest tests/test_bgemm_gpu.py -x -q -k "linear or conv" > $O/pytest_tile2.txt 2>&1 || { tail -40 $O/pytest_tile2.txt; exit 1; }
tail -3 $O/pytest_tile2.txt
REPS=20 python tools/bench_bgemm.py 2>/dev/null | cut -c1-42 > $O/bgemm_default.txt
DS6G_BG_TILE=2 REPS=20 python tools/bench_bgemm.py 2>/dev/null | cut -c1-42 > $O/bgemm_tile2.txt
paste $O/bgemm_default.txt $O/bgemm_tile2.txt
DS6G_BG_TILE=2 DS6G_LIB=deepsense6g_tii_amd/libds6g_gemmclk.so WHICH=bf16 timeout -k 10 300 python tools/gemm_clocks.py > $O/gemm_clocks_tile2.txt 2>&1
grep -A2 "bf16 linear fwd\|bf16 linear dgrad" $O/gemm_clocks_tile2.txt | grep -v "^--" | cut -c1-330 | head -24
