cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04h; mkdir -p $O
DS6G_LIB=deepsense6g_tii_amd/libds6g_gemmclk.so WHICH=f32,bf16,conv timeout -k 10 300 python tools/gemm_clocks.py > $O/gemm_clocks.txt 2>&1 || { tail -20 $O/gemm_clocks.txt; exit 1; }
cat $O/gemm_clocks.txt
