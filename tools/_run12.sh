cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04l; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_dp_gpu.py -x -q -s -k "resident_channel" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
grep "winograd_pc_kernel\|passed" $O/pytest.txt
for r in 0 16 32; do DS6G_PC_CU_RESERVE=$r timeout -k 10 200 python tools/coresidency.py 2>/dev/null | tail -1 >> $O/coresidency.jsonl || exit 1; done
cat $O/coresidency.jsonl
